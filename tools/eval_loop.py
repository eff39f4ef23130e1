"""The per-view body of the reference's evaluation loop (eval_nvidia.py:305-400) on Balloon1-shaped synthetic data: ray sampler for the
whole frame, the four feature-encoder passes (coarse + fine nets on the 7 dynamic and the 11 static source images), render_single_image_nvi
(64 coarse + 64 fine samples, chunk 8192), the pixels to the host and the masked PSNR the script computes there.  SSIM / LPIPS need the
script's external packages (skimage, the LPIPS network) and stay with the script.  Shared by bench.py (extra.eval_loop) and runnable alone:
    python tools/eval_loop.py [views]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynibar_amd import feature_network, synthetic as syn  # noqa: E402
from frame_case import FrameCase  # noqa: E402


class EvalLoop:
  def __init__(self, dev):
    self.fc = FrameCase(dev)
    self.dev = dev
    self.enc = feature_network.ResNet.from_module(syn.make_encoder_weights(0))
    self.enc_fine = feature_network.ResNet.from_module(syn.make_encoder_weights(1))
    g = torch.Generator().manual_seed(5)
    self.gt = torch.rand(self.fc.H, self.fc.W, 3, generator=g).numpy()

  def one_view(self):
    """-> (timings in ms, psnr) of one target view, following eval_nvidia.py:318-400"""
    fc, dev = self.fc, self.dev
    sync = lambda: torch.cuda.synchronize(dev)
    t = {}
    sync(); t0 = time.perf_counter()
    smp, rb = fc.sampler()                                    # RaySamplerSingleImage(data).get_all()  (:332-333)
    sync(); t['sampler'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
      src = rb['src_rgbs'].squeeze(0).permute(0, 3, 1, 2)          # (:335-358): four encoder passes
      st = rb['static_src_rgbs'].squeeze(0).permute(0, 3, 1, 2)
      ref_fm, _ = self.enc(src)
      _, st_fm = self.enc(st)
      ref_fm_f, _ = self.enc_fine(src)
      _, st_fm_f = self.enc_fine(st)
      sync(); t['encoders'] = time.perf_counter() - t0
      t0 = time.perf_counter()
      fc.cfeat, fc.ffeat = (ref_fm, None, st_fm), (ref_fm_f, None, st_fm_f)
      ret = fc.render(smp, rb)                                     # render_single_image_nvi (:360-378)
      sync(); t['render'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    rgb = ret['outputs_fine_ref']['rgb'].detach().cpu().numpy()   # (:380-381)
    depth = ret['outputs_fine_ref']['depth'].detach().cpu().numpy()
    valid = np.tile(np.float32(np.sum(rgb, axis=-1, keepdims=True) > 1e-3), (1, 1, 3))
    gt, pred = self.gt * valid, rgb * valid
    mse = float(np.sum((gt - pred) ** 2 * valid) / (np.sum(valid) + 1e-8))   # calculate_psnr with a mask (eval_nvidia.py:62-79)
    psnr = -10.0 * np.log10(mse + 1e-12)
    t['to_host_and_psnr'] = time.perf_counter() - t0
    assert depth.shape == rgb.shape[:2]
    return {k: v * 1e3 for k, v in t.items()}, psnr


def run(dev='cuda:0', views=2):
  loop = EvalLoop(dev)
  loop.one_view()  # warm-up: packs the networks and the encoders
  acc, ps = {}, []
  for _ in range(views):
    t, p = loop.one_view()
    ps.append(p)
    for k, v in t.items():
      acc[k] = acc.get(k, 0.0) + v / views
  total = sum(acc.values())
  return {'what': "per-view body of eval_nvidia.py's loop (:318-400) on Balloon1-shaped synthetic data: sampler, 4 encoder passes (7 + 11 images, coarse + "
                  'fine nets), render_single_image_nvi (288x512, 64 + 64 samples, 7 + 11 views), pixels to the host, masked PSNR',
          'ms_per_view': total, 'ms': {k: round(v, 2) for k, v in acc.items()}, 'views_timed': views,
          'balloon1_views': '(num_frames - 6) x 11 target views per scene (eval_nvidia.py:305-316)',
          'balloon1_gpu_path_minutes_at_24_frames': 18 * 11 * total / 6e4,
          'scope': 'GPU-PATH TIME ON SYNTHETIC DATA with random weights: what the loop does on the device + the pixel copy.  Not an evaluation wall-clock: image '
                   'decoding, SSIM / LPIPS (eval_nvidia.py:383-400) and checkpoint loading are outside it, and no trained weights or dataset are available offline',
          'psnr_vs_random_target_db': float(np.mean(ps))}


if __name__ == '__main__':
  print(json.dumps(run(views=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
