cd /root/repo; O=gpurun_out
echo "== framebench (new process)"; python tools/framebench.py --frames 1 2>&1 | grep "frame 1"
echo "== eval_loop (new process)"; python tools/eval_loop.py 1 2>&1 | tail -1 | cut -c1-400
echo "== framebench again (new process)"; python tools/framebench.py --frames 1 2>&1 | grep "frame 1"
echo "== one process: frame, then encoders only, then frame"
python - <<'PY'
import sys, time, torch
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from frame_case import FrameCase
from dynibar_amd import feature_network, synthetic as syn
fc = FrameCase('cuda:0'); smp, rb = fc.sampler(); fc.render(smp, rb); torch.cuda.synchronize()
def frame(tag):
  t0 = time.perf_counter(); fc.render(smp, rb); torch.cuda.synchronize(); print(tag, round((time.perf_counter() - t0) * 1e3, 1), 'ms', flush=True)
frame('frame before encoders')
enc = feature_network.ResNet.from_module(syn.make_encoder_weights(0))
src = rb['src_rgbs'].squeeze(0).permute(0, 3, 1, 2)
with torch.no_grad():
  a, b = enc(src)
torch.cuda.synchronize()
frame('frame after running the encoder (synthetic maps still)')
fc.cfeat = (a, None, fc.cfeat[2])
frame('frame with the encoder output as dynamic coarse maps')
PY
