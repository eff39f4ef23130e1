"""Hot-path bench: rays/sec of the per-ray renderer on synthetic Balloon1-shaped inputs (BASELINE.json configs[1]).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the static-branch hot path over one batch of R = 4096 rays per GPU (64 coarse samples, 8 source
views): sample_along_ray -> project_gather -> DynibarStatic (views / points / blend kernels) -> composite,
all through the C-ABI of libdynibar_hip.so with inputs resident in HBM.  With N > 1 every rank renders its own tile of rays
(weak scaling: rays are independent) and each step ends with the RCCL all-gather of the rendered pixels.
Prints ONE JSON line (rank 0).  The CPU leg times the oracle (the reference algorithm restated on torch-CPU) on a bounded
sample of the same workload; it is a reported baseline, not the target.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, F = 288, 512, 32
# algorithmic FLOPs (2 x MAC of the reference's Linear layers, mlp_network.py:333-405) per point-view of k_static_views:
# ray_dir_fc 103x256+256x35, base_fc 210x256+256x128, vis_fc 128x128+128x129, vis_fc2 128x128+128x1
FLOP_VIEWS_PER_PV = 2 * (103 * 256 + 256 * 35 + 210 * 256 + 256 * 128 + 128 * 128 + 128 * 129 + 128 * 128 + 128)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
# The network kernels run fp32-accurate products on the bf16 matrix pipe: every fp32 operand is split exactly into three bf16 parts and a
# product keeps 3 (default build) or 6 (fp32-class build) partial products (csrc/dyn_mlp.h), so the matrix-pipe ceiling for ALGORITHMIC
# fp32 FLOPs is the bf16 peak / terms.
def split_peak(terms):
  """matrix-pipe ceiling for ALGORITHMIC fp32 FLOPs: bf16 dense peak / partial products kept (0 = native fp32 MFMA engine)."""
  return FP32_MFMA_PEAK_TFLOPS if terms == 0 else BF16_MFMA_PEAK_TFLOPS / terms


def static_net_flops_per_point(S, V):
  """Linear layers + attention matmuls of DynibarStatic, per sample point."""
  return 0.361e6 + 0.033e6 * (S / 64.0) + 0.4305e6 * V


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--rays', type=int, default=4096, help='rays per step per GPU (N_rand / chunk)')
  ap.add_argument('--samples', type=int, default=64)
  ap.add_argument('--views', type=int, default=8)
  ap.add_argument('--cpu-rays', type=int, default=256, help='rays of the same workload timed on the host oracle (0 = skip)')
  ap.add_argument('--no-x6', action='store_true', help='skip the extra leg that times the fp32-class (6-term split) engine build')
  a = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if a.gpus > 1 and world != a.gpus:
    raise SystemExit(f'--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})')
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
  dev = torch.device('cuda', local_rank)
  torch.cuda.set_device(dev)
  dist = None
  if world > 1:
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=dev)

  from dynibar_amd import _lib, ops, synthetic as syn
  lib = _lib.lib()

  R, S, V = a.rays, a.samples, a.views
  sc = syn.make_scene(seed=0, H=H, W=W, V=V, F=F, n_static=V)
  T = lambda x: torch.from_numpy(x).to(dev)
  scene = {k: T(v) for k, v in sc.items()}
  pix = syn.sample_pixels(100 + rank, H, W, R)  # each rank renders its own tile of rays
  o_np, d_np, _ = syn.pixel_rays(sc['camera'], pix)
  ray_o, ray_d = T(o_np), T(d_np)
  weights = syn.make_weights('static', 0, F)
  net = ops.StaticNet(weights, dev, anti_alias_pooling=True, mask_rgb=False)
  views = ops.SourceViews(scene['camera'], scene['static_src_rgbs'], scene['static_src_cameras'], scene['static_featmaps'])
  gathered = torch.empty((world * R, 4), dtype=torch.float32, device=dev) if world > 1 else None

  def step():
    pts, z, _ = ops.sample_along_ray(ray_o, ray_d, scene['depth_range'], S, True, want_s=False)
    rgb_feat, ray_diff, mask, pm = ops.project_gather(views, R, S, ray_o=ray_o, ray_d=ray_d, z_vals=z, pix_mask_thresh=1.0)
    raw = net(views, ray_o, ray_d, pts, rgb_feat, ray_diff, mask)
    out = ops.composite(raw, z, pm, per_sample=False)
    if world > 1:
      dist.all_gather_into_tensor(gathered, torch.cat([out['rgb'], out['depth'][:, None]], dim=1))
    return out

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize()

  for _ in range(a.warmup):
    out = step()
  fence()
  lib.dyn_profile_enable(1)
  t0 = time.perf_counter()
  for _ in range(a.steps):
    out = step()
  fence()
  dt = time.perf_counter() - t0
  lib.dyn_profile_enable(0)
  if world > 1:
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

  nk = lib.dyn_profile_count()
  ms = (ctypes.c_float * nk)()
  cnt = (ctypes.c_int * nk)()
  lib.dyn_profile_read(ms, cnt)
  kernels = {lib.dyn_profile_name(i).decode(): {'launches': cnt[i], 'avg_ms': ms[i] / cnt[i]} for i in range(nk) if cnt[i] > 0}

  if rank != 0:
    if world > 1:
      dist.destroy_process_group()
    return

  value = world * R * a.steps / dt
  terms = int(lib.dyn_mlp_split_terms())
  B6_PEAK_TFLOPS = split_peak(terms)
  dom = kernels['k_static_views']
  flops_launch = FLOP_VIEWS_PER_PV * R * S * V
  achieved = flops_launch / (dom['avg_ms'] * 1e-3) / 1e12
  net_ms = sum(kernels[k]['avg_ms'] for k in ('k_static_ref_feat', 'k_static_views', 'k_static_points', 'k_static_blend'))
  net_tflops = static_net_flops_per_point(S, V) * R * S / (net_ms * 1e-3) / 1e12
  # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process; they are collected with
  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over this same command and stored by tools/rocpd_summary.py traffic
  traffic, traffic_note = None, 'no PMC summary for this build (profiles/r01_traffic.json)'
  tpath = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
  if os.path.exists(tpath) and (R, S, V) == (4096, 64, 8) and terms == 3:
    try:
      with open(tpath) as f:
        tj = json.load(f)['k_static_views']
      # gfx950 rocprofv3: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section)
      traffic = 2 * tj['FETCH_SIZE_KB'] * 1024 + tj['WRITE_SIZE_KB'] * 1024
      traffic_note = (f"bytes per launch from {tj['source']}: 2 x FETCH_SIZE ({tj['FETCH_SIZE_KB']:.0f} KB, gfx950 half-count correction) + "
                      f"WRITE_SIZE ({tj['WRITE_SIZE_KB']:.0f} KB); collected on the same bench command, not in this run")
    except Exception as e:
      traffic_note = f'profiles/r01_traffic.json unreadable: {e}'
  pg = kernels['k_project_gather']
  pg_bytes = R * S * V * 160 + V * ((H // 4) * (W // 4) * F + H * W * 3) * 4 + R * (24 + 4 * S)
  res = {
      'metric': 'rays/sec (64 samples x 8 src views)', 'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': a.steps,
      'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32' if terms == 0 else f'f32 (bf16x{terms} split-product MFMA, f32 accumulate)', 'data': 'synthetic',
      'config': {'workload': 'BASELINE configs[1]: Nvidia Balloon1 eval shape, static branch only '
                             '(sample -> project/gather -> DynibarStatic -> composite)',
                 'rays_per_step_per_gpu': R, 'samples': S, 'src_views': V, 'src_image': [H, W], 'feature_map': [F, H // 4, W // 4],
                 'sharding': 'ray tiles per rank + RCCL all-gather of rendered pixels' if world > 1 else 'single GPU'},
      'roofline': {'kernel': 'k_static_views', 'bound': 'mfma', 'achieved': achieved, 'peak': B6_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': achieved / B6_PEAK_TFLOPS, 'traffic': traffic, 'traffic_note': traffic_note, 'avg_launch_ms': dom['avg_ms'],
                   'algorithmic_flops_per_launch': flops_launch,
                   'peak_note': f'fp32 operands as exact bf16 splits, {terms} partial products per product on the bf16 matrix pipe, fp32 '
                                f'accumulation: peak = 2500 TFLOP/s dense bf16 MFMA / {terms}; the native fp32 MFMA peak is 157.3 TFLOP/s',
                   'vs_fp32_mfma_peak': achieved / FP32_MFMA_PEAK_TFLOPS},
      'roofline_static_net': {'bound': 'mfma', 'achieved': net_tflops, 'peak': B6_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': net_tflops / B6_PEAK_TFLOPS, 'avg_ms': net_ms, 'vs_fp32_mfma_peak': net_tflops / FP32_MFMA_PEAK_TFLOPS},
      'roofline_project_gather': {'bound': 'hbm', 'achieved': pg_bytes / (pg['avg_ms'] * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                                  'frac': pg_bytes / (pg['avg_ms'] * 1e-3) / 8e12, 'avg_launch_ms': pg['avg_ms'],
                                  'algorithmic_bytes_per_launch': pg_bytes},
      'kernels_avg_ms': {k: round(v['avg_ms'], 5) for k, v in kernels.items()},
  }

  if a.cpu_rays > 0 and world == 1:
    # the oracle (test infrastructure) is used here ONLY as the timed CPU baseline and as the checker of this run's pixels
    from oracle import ibr_oracle as O
    n = min(a.cpu_rays, R)
    cpu_scene = {k: torch.from_numpy(v) for k, v in sc.items()}
    sd = O.tdict(weights)
    co, cd = torch.from_numpy(o_np[:n]), torch.from_numpy(d_np[:n])
    # torch-CPU oversubscribes on these small per-ray tensors: pick the best thread count on a 32-ray probe, then time the sample
    best = None
    with torch.no_grad():
      for th in sorted({8, 32, min(64, os.cpu_count() or 1), os.cpu_count() or 1}):
        torch.set_num_threads(th)
        O.static_branch_pass(sd, cpu_scene, co[:32], cd[:32], S, True, True)
        t1 = time.perf_counter()
        O.static_branch_pass(sd, cpu_scene, co[:32], cd[:32], S, True, True)
        tp = time.perf_counter() - t1
        if best is None or tp < best[1]:
          best = (th, tp)
      cores = best[0]
      torch.set_num_threads(cores)
      t1 = time.perf_counter()
      ref = O.static_branch_pass(sd, cpu_scene, co, cd, S, True, True)
      cpu_dt = time.perf_counter() - t1
    res['cpu_baseline'] = {'value': n / cpu_dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
                           'sample': f'{n} of the {R} rays of one step, same scene/weights, torch-CPU oracle, {cores} threads (best of 8/32/64/all on a 32-ray probe; host has {os.cpu_count()} hardware threads), 1 run after warm-up'}
    err = (out['rgb'][:n].cpu() - ref['rgb']).abs()
    mse = float((err ** 2).mean())
    res['check_vs_oracle'] = {'rays': n, 'max_abs_rgb_err': float(err.max()), 'psnr_db': (10 * np.log10(1.0 / mse)) if mse > 0 else float('inf')}
  x6 = os.path.join(ROOT, 'dynibar_amd', 'csrc', 'libdynibar_hip_x6.so')
  if world == 1 and not a.no_x6 and terms == 3 and os.path.exists(x6) and not os.environ.get('DYNIBAR_HIP_LIB'):
    # the same bench on the fp32-class build (6 partial products), as a second reported number
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', str(max(5, a.steps // 2)), '--warmup', '2', '--cpu-rays', '0', '--no-x6',
                        '--rays', str(R), '--samples', str(S), '--views', str(V)], env=dict(os.environ, DYNIBAR_HIP_LIB=x6),
                       capture_output=True, text=True, timeout=600)
    try:
      d6 = json.loads(r.stdout.strip().splitlines()[-1])
      res['fp32_class_engine'] = {'value': d6['value'], 'unit': 'rays/s', 'ms_per_step': d6['ms_per_step'], 'dtype': d6['dtype'],
                                  'k_static_views_ms': d6['kernels_avg_ms']['k_static_views']}
    except Exception as e:  # the extra leg must never cost the main line
      res['fp32_class_engine'] = {'error': str(e)[:200]}
  print(json.dumps(res))
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
