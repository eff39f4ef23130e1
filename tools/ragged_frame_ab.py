"""Developer check: one full Nvidia-eval frame (tools/frame_case.py) rendered in separate processes under different settings (ragged rows on / off, one or two chunk
streams), twice each; the rendered coarse / fine colours of ALL rays are compared pairwise.  python tools/ragged_frame_ab.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tools'))
from frame_case import FrameCase
fc = FrameCase('cuda:0')
smp, rb = fc.sampler()
outs = []
for rep in range(2):
  ret = fc.render(smp, rb)
  torch.cuda.synchronize()
  outs.append(np.concatenate([ret['outputs_coarse_ref']['rgb'].reshape(-1, 3).cpu().numpy(), ret['outputs_fine_ref']['rgb'].reshape(-1, 3).cpu().numpy()], 1))
print('same process, frame 0 vs frame 1: max diff', float(np.abs(outs[0] - outs[1]).max()))
np.save(OUT, outs[1])
'''
def main():
  runs = [('rag_s2', dict(DYN_RAGGED='1', DYNIBAR_CHUNK_STREAMS='2')), ('rag_s2b', dict(DYN_RAGGED='1', DYNIBAR_CHUNK_STREAMS='2')),
          ('rag_s1', dict(DYN_RAGGED='1', DYNIBAR_CHUNK_STREAMS='1')), ('reg_s2', dict(DYN_RAGGED='0', DYNIBAR_CHUNK_STREAMS='2')),
          ('reg_s1', dict(DYN_RAGGED='0', DYNIBAR_CHUNK_STREAMS='1'))]
  if '--many' in sys.argv:
    runs = [(f'rag_s2_{i}', dict(DYN_RAGGED='1', DYNIBAR_CHUNK_STREAMS='2')) for i in range(6)] + [('rag_s1', dict(DYN_RAGGED='1', DYNIBAR_CHUNK_STREAMS='1'))]
  if '--quick' in sys.argv:
    runs = [r for r in runs if r[0] in ('rag_s2', 'rag_s2b', 'rag_s1')] + [('rag_s3', dict(DYN_RAGGED='1', DYNIBAR_CHUNK_STREAMS='3'))]
  res = {}
  for tag, env in runs:
    out = f'/tmp/ragged_frame_{tag}.npy'
    code = (CHILD % (ROOT, ROOT)).replace('OUT', repr(out))
    pr = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), capture_output=True, text=True)
    print(tag, env, '|', pr.stdout.strip()[-120:], pr.stderr.strip()[-500:] if pr.returncode else '', flush=True)
    if pr.returncode == 0:
      res[tag] = np.load(out)
  tags = list(res)
  for i in range(len(tags)):
    for j in range(i + 1, len(tags)):
      if '--many' in sys.argv and tags[j] != 'rag_s1':
        continue
      d = np.abs(res[tags[i]] - res[tags[j]])
      dc, df = d[:, :3].max(1), d[:, 3:].max(1)
      print(f'{tags[i]:8s} vs {tags[j]:8s}: coarse max {dc.max():.3e} rays > 2e-5: {int((dc > 2e-5).sum()):6d}   fine max {df.max():.3e} rays > 2e-5: {int((df > 2e-5).sum()):6d}'
            + (f'   worst coarse ray {int(dc.argmax())} (chunk {int(dc.argmax()) // 8192}, pixel row {int(dc.argmax()) // 512})' if dc.max() > 2e-5 else ''), flush=True)
if __name__ == '__main__':
  main()
