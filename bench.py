"""Hot-path bench: rays/sec of the per-ray renderer on synthetic Balloon1-shaped inputs (BASELINE.json configs[1]).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the static-branch hot path over one batch of R = 4096 rays per GPU (64 coarse samples, 8 source
views): sample_along_ray -> project_gather -> DynibarStatic (views / points / blend kernels) -> composite, all through the
C-ABI of libdynibar_hip.so with inputs resident in HBM.  With N > 1 every rank renders its own tile of rays (weak scaling: rays
are independent) and each step ends with the RCCL all-gather of the rendered pixels.
Prints ONE JSON line (rank 0).  The CPU leg times the oracle (the reference algorithm restated on torch-CPU) on a bounded
sample of the same workload (median of 3); it is a reported baseline, not the target.  Extra legs in the same line (`extra`): the
same step at the Nvidia eval's real static view count (11), one full 288x512 frame through render_single_image_nvi (on N > 1 GPUs:
ray-tiled across the ranks -- a strong-scaling number next to the weak-scaling headline), and the HBM traffic of the network and
gather kernels measured by rocprofv3 --pmc children of this same command.  With N > 1 the line also carries every rank's own ms per step, the
cost of the pixel all-gather alone and, for the frame leg, every rank's render and gather times.
`--dry-run` rehearses exactly this control flow on CPU ranks over gloo with a stub of the kernel layer (tests/test_bench_dry_run.py launches it
through torch.distributed.run with 2 ranks): the product has no CPU path, so the numbers of a dry run mean nothing -- the point is that the
N > 1 command cannot fail on first contact with an 8-GPU node for a reason a CPU could have found.
"""
import argparse
import ctypes
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

H, W, F = 288, 512, 32
# algorithmic FLOPs (2 x MAC of the reference's Linear layers, mlp_network.py:333-405) per point-view of k_static_views:
# ray_dir_fc 103x256+256x35, base_fc 210x256+256x128, vis_fc 128x128+128x129, vis_fc2 128x128+128x1
FLOP_VIEWS_PER_PV = 2 * (103 * 256 + 256 * 35 + 210 * 256 + 256 * 128 + 128 * 128 + 128 * 129 + 128 * 128 + 128)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
HALF_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA (v_mfma_f32_32x32x16_{bf16,f16})
HBM_PEAK_GBPS = 8000.0
KIND = {0: 'native fp32 MFMA', 1: 'bf16', 2: 'f16'}


def split_peak(terms):
  """Matrix-pipe ceiling for ALGORITHMIC fp32 FLOPs: the network kernels split every fp32 operand into 16-bit parts (two half floats = 22 mantissa bits
  in the shipped engine, three bf16 = all 24 in the 6-term build) and keep `terms` partial products per product on the 16-bit matrix pipe (csrc/dyn_mlp.h), so the ceiling is the dense 16-bit peak / terms
  (0 = the native fp32 MFMA engine)."""
  return FP32_MFMA_PEAK_TFLOPS if terms == 0 else HALF_MFMA_PEAK_TFLOPS / terms


def static_net_flops_per_point(S, V):
  """Linear layers + attention matmuls of DynibarStatic, per sample point."""
  return 0.361e6 + 0.033e6 * (S / 64.0) + 0.4305e6 * V


def gather_bytes(R, S, V):
  """compulsory bytes of the fused projection/gather per launch."""
  return R * S * V * 160 + V * ((H // 4) * (W // 4) * F + H * W * 3) * 4 + R * (24 + 4 * S)


def read_kernels(lib):
  nk = lib.dyn_profile_count()
  ms = (ctypes.c_float * nk)()
  cnt = (ctypes.c_int * nk)()
  lib.dyn_profile_read(ms, cnt)
  return {lib.dyn_profile_name(i).decode(): {'launches': cnt[i], 'avg_ms': ms[i] / cnt[i]} for i in range(nk) if cnt[i] > 0}


class StaticStep:
  """The bench workload on one device: everything resident in HBM; step() launches one pass of the hot path."""

  def __init__(self, dev, R, S, V, rank=0):
    from dynibar_amd import ops, synthetic as syn
    self.ops, self.R, self.S, self.V = ops, R, S, V
    self.sc = syn.make_scene(seed=0, H=H, W=W, V=V, F=F, n_static=V)
    T = lambda x: torch.from_numpy(x).to(dev)
    self.scene = {k: T(v) for k, v in self.sc.items()}
    pix = syn.sample_pixels(100 + rank, H, W, R)  # each rank renders its own tile of rays
    self.o_np, self.d_np, _ = syn.pixel_rays(self.sc['camera'], pix)
    self.ray_o, self.ray_d = T(self.o_np), T(self.d_np)
    self.weights = syn.make_weights('static', 0, F)
    self.net = ops.StaticNet(self.weights, dev, anti_alias_pooling=True, mask_rgb=False)
    self.views = ops.SourceViews(self.scene['camera'], self.scene['static_src_rgbs'], self.scene['static_src_cameras'], self.scene['static_featmaps'])

  def step(self):
    ops, R, S = self.ops, self.R, self.S
    pts, z, _ = ops.sample_along_ray(self.ray_o, self.ray_d, self.scene['depth_range'], S, True, want_s=False)
    rgb_feat, ray_diff, mask, pm = ops.project_gather(self.views, R, S, ray_o=self.ray_o, ray_d=self.ray_d, z_vals=z, pix_mask_thresh=1.0)
    raw = self.net(self.views, self.ray_o, self.ray_d, pts, rgb_feat, ray_diff, mask)
    return ops.composite(raw, z, pm, per_sample=False)


class DryStep:
  """--dry-run: the shape of StaticStep without kernels (a per-ray function on the CPU); bench-only, never part of the product."""

  def __init__(self, dev, R, S, V, rank=0):
    g = torch.Generator().manual_seed(100 + rank)
    self.R, self.ray_o = R, torch.rand(R, 3, generator=g)

  def step(self):
    return {'rgb': self.ray_o * 0.5 + 0.25, 'depth': self.ray_o.sum(dim=1)}


def dry_render_rays_mv(frame_idx, time_embedding, time_offset, ray_batch, model, projector, coarse_featmaps, fine_featmaps, N_samples, args,
                       inv_uniform=False, N_importance=0, raw_noise_std=0.0, det=False, white_bkgd=False, is_train=True):
  """--dry-run: render_rays_mv's output structure from a per-ray function (what tests/test_distributed_cpu.py drives the frame code with)."""
  o = ray_batch['ray_o']
  s = torch.arange(N_samples, dtype=torch.float32)[None, :]
  coarse = {'rgb': o * 2.0 + 1.0, 'depth': o.sum(dim=1), 'weights': o[:, :1] * s, 'mask': o[:, 0] > 0.3}
  fine = {'rgb': o * 3.0, 'mask': o[:, 1] > 0.5, 'depth': o[:, 2], 'alpha': o[:, 1:2] * s}
  return {'outputs_coarse_ref': coarse, 'outputs_fine_ref': fine, 'outputs_fine_anchor': None, 'outputs_fine_anchor_dy': None}


class DryFrameCase:
  """--dry-run: FrameCase's interface; render() is the REAL render_single_image_nvi (ray tiles, chunk slicing, packed all-gather) over the stub."""

  def __init__(self, dev, H=36, W=64, chunk=512):
    import types
    from dynibar_amd import render_image
    self.RI, self.H, self.W, self.chunk = render_image, H, W, chunk
    render_image.render_rays_mv = dry_render_rays_mv
    self.args = types.SimpleNamespace(frame_outputs=None)

  def sampler(self):
    import types
    g = torch.Generator().manual_seed(5)
    return types.SimpleNamespace(H=self.H, W=self.W), {'ray_o': torch.rand(self.H * self.W, 3, generator=g), 'camera': torch.zeros(1, 34), 'rgb': None}

  def render(self, smp, rb):
    return self.RI.render_single_image_nvi((0, None), (None, None), ([0], None), smp, rb, None, None, self.chunk, 4, self.args, N_importance=4, det=True,
                                           is_train=False)


def timed(lib, step, steps, warmup, fence):
  for _ in range(warmup):
    out = step()
  fence()
  if lib is not None:
    lib.dyn_profile_enable(1)
  t0 = time.perf_counter()
  for _ in range(steps):
    out = step()
  fence()
  dt = time.perf_counter() - t0
  if lib is None:
    return out, dt, {}
  lib.dyn_profile_enable(0)
  return out, dt, read_kernels(lib)


def pmc_clock(a):
  """Effective shader clock and matrix-pipe occupancy of k_static_views from one more rocprofv3 --pmc child (GRBM_GUI_ACTIVE is summed over the 8
  XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles: 32 per v_mfma_f32_32x32x16).  The dense MFMA peak of the guide is quoted at 2.4 GHz; the chip
  clocks to its power budget (MI355X_MICROARCH.md, DVFS), so the fraction at the clock the kernel actually ran at is reported next to the nominal one."""
  got, note = pmc_traffic(a, counters=('GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES',), raw=True)
  if not got or 'k_static_views' not in got:
    return None, note
  v = got['k_static_views']
  return {'grbm_gui_active': v.get('GRBM_GUI_ACTIVE'), 'mfma_busy_cycles': v.get('SQ_VALU_MFMA_BUSY_CYCLES'), 'avg_us_under_profiler': v.get('avg_us')}, note


def pmc_insts(a):
  """Instruction mix of the network kernels from one more --pmc child: SQ_INSTS_VALU counts every VALU-class instruction a wave issued (MFMAs included),
  SQ_INSTS_MFMA the matrix instructions, SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE how much of the launch the shader engines had waves resident (the slow-state
  diagnostic of the one-workgroup-per-CU kernels: DESIGN.md section 5)."""
  got, note = pmc_traffic(a, counters=('SQ_INSTS_VALU SQ_INSTS_MFMA', 'SQ_BUSY_CYCLES GRBM_GUI_ACTIVE'), raw=True)
  return got, note


def pmc_traffic(a, counters=('FETCH_SIZE', 'WRITE_SIZE'), raw=False):
  """HBM-side bytes per launch of every kernel of this bench, by rocprofv3 --pmc children of this same command (one counter per pass, as
  MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950: wide coalesced reads are tallied at half their bytes)."""
  import glob
  import shutil
  import sqlite3
  import tempfile
  exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
  if not os.path.exists(exe):
    return None, 'rocprofv3 not found'
  res = {}
  for c in counters:
    d = tempfile.mkdtemp(prefix='dynibar_pmc_', dir=os.environ.get('TMPDIR', '/tmp'))
    try:
      cmd = [exe, '--kernel-trace', '--pmc'] + c.split() + ['-d', d, '--', sys.executable, os.path.abspath(__file__), '--child', '--steps', '4', '--warmup', '1',
             '--rays', str(a.rays), '--samples', str(a.samples), '--views', str(a.views)]
      r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=d, env=dict(os.environ, TMPDIR=d))
      dbs = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
      if not dbs:
        return None, f'rocprofv3 --pmc {c}: no database written (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}'
      cur = sqlite3.connect(dbs[0]).cursor()
      for cn in c.split():
        for name, n, v in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (cn,)):
          key = name.replace('void ', '').split('<')[0].split('(')[0]
          res.setdefault(key, {})[cn if raw else cn + '_KB'] = v
          res[key]['dispatches'] = n
      if raw:
        try:  # counters_collection carries the dispatch's duration (ns) next to every counter value
          for name, ns in cur.execute("select kernel_name, avg(duration) from counters_collection where counter_name = ? group by kernel_name", (c.split()[0],)):
            key = name.replace('void ', '').split('<')[0].split('(')[0]
            if key in res:
              res[key]['avg_us'] = ns / 1000.0
        except Exception:
          pass
    except Exception as e:  # the profiler leg must never cost the main line
      return None, f'rocprofv3 --pmc {c} failed: {str(e)[:200]}'
    finally:
      shutil.rmtree(d, ignore_errors=True)
  if raw:
    return res, 'rocprofv3 --kernel-trace --pmc ' + ' '.join(counters) + ' child of this command (4 steps)'
  out = {}
  for k, v in res.items():
    if k.startswith('k_') and 'FETCH_SIZE_KB' in v and 'WRITE_SIZE_KB' in v:
      out[k] = {'bytes': 2 * v['FETCH_SIZE_KB'] * 1024 + v['WRITE_SIZE_KB'] * 1024, 'fetch_kb_raw': v['FETCH_SIZE_KB'], 'write_kb': v['WRITE_SIZE_KB'],
                'dispatches': v['dispatches']}
  return out, ('bytes per launch = 2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE, rocprofv3 --kernel-trace --pmc <one counter per pass> '
               'children of this command (4 steps each)')


def power_sample(step, sync, seconds=4.0):
  """rocm-smi readings (package power, shader clock, power cap) taken by a side thread while `step` runs back to back for `seconds`."""
  import re
  import subprocess
  import threading
  smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
  samples = []

  def sampler():
    time.sleep(1.0)
    for _ in range(3):
      try:
        out = subprocess.run([smi, '--showpower', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
        w = re.search(r'Package Power \(W\):\s*([0-9.]+)', out)
        c = re.search(r'sclk clock level:\s*\d+:\s*\((\d+)Mhz\)', out)
        if w and c:
          samples.append((float(w.group(1)), int(c.group(1))))
      except Exception:
        pass
      time.sleep(0.3)

  th = threading.Thread(target=sampler)
  th.start()
  t0 = time.perf_counter()
  n = 0
  while time.perf_counter() - t0 < seconds or th.is_alive():
    for _ in range(50):
      step()
    sync()
    n += 50
  dt = time.perf_counter() - t0
  th.join()
  cap = None
  try:
    out = subprocess.run([smi, '--showmaxpower'], capture_output=True, text=True, timeout=10).stdout
    m = re.search(r'Max Graphics Package Power \(W\):\s*([0-9.]+)', out)
    cap = float(m.group(1)) if m else None
  except Exception:
    pass
  if not samples:
    return {'error': 'rocm-smi gave no reading'}
  return {'what': 'rocm-smi beside %d back-to-back steps (%.2f ms per step)' % (n, dt / n * 1e3), 'package_power_w': [s[0] for s in samples],
          'shader_clock_mhz': [s[1] for s in samples], 'power_cap_w': cap, 'nominal_clock_mhz': 2400}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--rays', type=int, default=4096, help='rays per step per GPU (N_rand / chunk)')
  ap.add_argument('--samples', type=int, default=64)
  ap.add_argument('--views', type=int, default=8)
  ap.add_argument('--cpu-rays', type=int, default=1024, help='rays of the same workload timed on the host oracle (0 = skip)')
  ap.add_argument('--no-x6', action='store_true', help='skip the leg that times the exact 6-term bf16 split engine (libdynibar_hip_x6.so) beside the shipped one')
  ap.add_argument('--x6', action='store_true', help='(kept for old command lines: the 6-term leg is on by default since round 6)')
  ap.add_argument('--no-extra', action='store_true', help='skip the extra legs (11 views, full frame)')
  ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 --pmc child runs that measure HBM traffic per launch')
  ap.add_argument('--dry-run', action='store_true', help='rehearse the (multi-rank) control flow on CPU ranks over gloo with a stub of the kernel layer; numbers are meaningless')
  ap.add_argument('--force-dist', action='store_true', help='with ONE rank under torch.distributed.run: initialise the process group (RCCL) anyway and take the '
                  'multi-rank code path -- per-step pixel all-gather, barrier-bracketed fences, the ray-tiled frame leg -- so that this path runs on a single MI355X')
  ap.add_argument('--gather', choices=('torch', 'abi'), default=None, help="the frame leg's pixel gather: torch.distributed (default) or the C-ABI's dyn_gather_tiles")
  ap.add_argument('--frame-only', action='store_true', help='of the extra legs only the full frame (tests)')
  ap.add_argument('--dump-rgb', default=None, help=argparse.SUPPRESS)  # (child of the x6 leg: the last step's colours of the first cpu-rays rays as .npy)
  ap.add_argument('--child', nargs='?', const='plain', default=None, help=argparse.SUPPRESS)  # 'plain': no baseline / extra legs; 'check': keep the oracle check (x6 leg)
  a = ap.parse_args()
  a.x6 = not a.no_x6
  if a.child:
    a.cpu_rays, a.no_extra, a.no_traffic, a.x6 = 0 if a.child == 'plain' else a.cpu_rays, True, True, False
  if a.dry_run:
    a.cpu_rays, a.no_traffic, a.x6 = 0, True, False

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if a.gpus > 1 and world != a.gpus:
    raise SystemExit(f'--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})')
  dry = a.dry_run
  if not dry and not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path; --dry-run only rehearses the control flow)')
  dev = torch.device('cpu') if dry else torch.device('cuda', local_rank)
  if not dry:
    torch.cuda.set_device(dev)
  dist = None
  multi_rank = world > 1 or a.force_dist  # the collective code path is live (also with a process group of one rank under --force-dist)
  if multi_rank:
    import torch.distributed as dist
    if 'MASTER_ADDR' not in os.environ:  # --force-dist without torch.distributed.run
      os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'), RANK='0', WORLD_SIZE='1')
    if dry:
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=dev)
    if a.force_dist:
      os.environ['DYNIBAR_FORCE_DIST'] = '1'
  if a.gather:
    os.environ['DYNIBAR_GATHER'] = a.gather

  lib = None
  if not dry:
    from dynibar_amd import _lib
    lib = _lib.lib()
  sync = (lambda: None) if dry else torch.cuda.synchronize

  # ---- N > 1 readiness, BEFORE anything is timed: does RCCL see all N ranks, through torch.distributed and through the package's own communicator? ----
  readiness = None
  if multi_rank:
    readiness = {'world_size': world, 'backend': dist.get_backend()}
    names = [None] * world
    dist.all_gather_object(names, 'cpu (dry run)' if dry else '%s #%d' % (torch.cuda.get_device_name(dev), local_rank))
    readiness['devices'] = names
    probe = torch.full((1,), float(rank), dtype=torch.float32, device=dev)
    got = torch.empty(world, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(got, probe)
    readiness['torch_allgather_ranks_seen'] = int((got == torch.arange(world, dtype=torch.float32, device=dev)).sum().item())
    assert readiness['torch_allgather_ranks_seen'] == world, f'all_gather saw {readiness["torch_allgather_ranks_seen"]} of {world} ranks'
    if not dry:
      try:
        from dynibar_amd import render_image as _ri
        from dynibar_amd._lib import call as _call, ptr as _ptr, stream_of as _stream_of
        comm = _ri.abi_communicator(dist, world, rank, dev)  # rank 0 draws the id, torch.distributed carries it, every rank joins (ncclCommInitRank)
        nr, rk = ctypes.c_int(0), ctypes.c_int(-1)
        _call('dyn_comm_size_rank', comm, ctypes.byref(nr), ctypes.byref(rk))  # ncclCommCount / ncclCommUserRank of that communicator
        readiness['rccl_ranks'], readiness['rccl_rank_of_rank0'] = int(nr.value), int(rk.value) if rank == 0 else None
        rows = 3 + rank  # (the real frame pads unequal tiles to the largest: the same shape here)
        tile = 3 + world - 1
        send = torch.zeros((tile, 5), dtype=torch.float32, device=dev)
        send[:rows] = float(rank + 1)
        recv = torch.empty((world * tile, 5), dtype=torch.float32, device=dev)
        _call('dyn_gather_tiles', _ptr(send), _ptr(recv), tile, 5, comm, _stream_of(send))
        sync()
        want = torch.zeros((world, tile, 5), dtype=torch.float32, device=dev)
        for r_ in range(world):
          want[r_, :3 + r_] = float(r_ + 1)
        readiness['dyn_gather_tiles_ok'] = bool(torch.equal(recv.view(world, tile, 5), want))
      except Exception as e:  # the package's own communicator is an option of the frame leg (--gather abi): its failure must not cost the headline
        readiness['rccl_ranks'] = None
        readiness['abi_communicator_error'] = str(e)[:300]
      seen = [None] * world
      dist.all_gather_object(seen, readiness.get('rccl_ranks'))
      readiness['rccl_ranks_every_rank'] = seen
      if readiness.get('rccl_ranks') is not None:
        assert readiness['rccl_ranks'] == world, f'the RCCL communicator has {readiness["rccl_ranks"]} ranks, expected {world}'

  R, S, V = a.rays, a.samples, a.views
  wl = (DryStep if dry else StaticStep)(dev, R, S, V, rank)
  gathered = torch.empty((world * R, 4), dtype=torch.float32, device=dev) if multi_rank else None

  def pixels(out):
    return torch.cat([out['rgb'], out['depth'][:, None]], dim=1)

  def step():
    out = wl.step()
    if multi_rank:
      dist.all_gather_into_tensor(gathered, pixels(out))
    return out

  def fence():
    sync()
    if multi_rank:
      dist.barrier()
      sync()

  def max_over_ranks(x):
    if not multi_rank:
      return x
    tt = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())

  def every_rank(x):
    """x of every rank, in rank order (a list of one on a single GPU)."""
    if not multi_rank:
      return [float(x)]
    tt = torch.zeros(world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(tt, torch.tensor([x], dtype=torch.float64, device=dev))
    return [float(v) for v in tt.tolist()]

  out, dt_own, kernels = timed(lib, step, a.steps, a.warmup, fence)
  dt = max_over_ranks(dt_own)
  multi = None
  if multi_rank:
    # what the collective alone costs (the same payload, the same stream, nothing to overlap with), and every rank's own clock
    send = pixels(out).contiguous()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
      dist.all_gather_into_tensor(gathered, send)
    fence()
    ag = max_over_ranks(time.perf_counter() - t0)
    multi = {'per_rank_ms_per_step': [round(v / a.steps * 1e3, 4) for v in every_rank(dt_own)], 'allgather_alone_ms_per_step': ag / a.steps * 1e3,
             'allgather_bytes_per_rank_per_step': int(send.numel() * 4), 'backend': dist.get_backend(), 'readiness': readiness,
             'rccl_ranks': (readiness or {}).get('rccl_ranks'), 'devices': (readiness or {}).get('devices')}

  # ---- extra legs (every rank takes part in the frame leg: the ray tiles are a collective effort) ----
  extra = {}
  if not a.no_extra:
    try:
      from dynibar_amd import render_image
      if dry:
        fc = DryFrameCase(dev)
      else:
        from frame_case import FrameCase
        fc = FrameCase(dev)
      smp, rb = fc.sampler()
      fc.render(smp, rb)  # warm-up: packs the six networks, prepares the source views
      fence()
      # three timed frames, the median counts (single frames of one session spread by several per cent, with an occasional +10 % outlier) ...
      frame_times = []
      for _ in range(1 if dry else 3):
        t0 = time.perf_counter()
        ret = fc.render(smp, rb)
        fence()
        frame_times.append(max_over_ranks(time.perf_counter() - t0))
      fdt = sorted(frame_times)[len(frame_times) // 2]
      # ... and one more for the opt-in stage clocks (they synchronise the device between the stages, so that frame is not the timed one)
      render_image.FRAME_STATS = {}
      ret = fc.render(smp, rb)
      fence()
      fst, render_image.FRAME_STATS = render_image.FRAME_STATS, None
      fk = {}
      if lib is not None:
        # the per-kernel breakdown comes from ONE MORE frame on a single stream: the timed frame above alternates its chunks over
        # render_image.CHUNK_STREAMS streams, and HIP-event brackets of overlapping kernels would count the overlap twice
        n_streams, render_image.CHUNK_STREAMS = render_image.CHUNK_STREAMS, 1
        from dynibar_amd import ops as _ops_mod
        _ops_mod.GATHER_STATS = {}
        lib.dyn_profile_enable(1)
        fc.render(smp, rb)
        fence()
        lib.dyn_profile_enable(0)
        render_image.CHUNK_STREAMS = n_streams
        fk = read_kernels(lib)
        frame_gather, _ops_mod.GATHER_STATS = _ops_mod.GATHER_STATS, None
      n_frame_rays = rb['ray_o'].shape[0]
      extra['frame_nvi_288x512'] = {
          'what': 'ONE render_single_image_nvi call (BASELINE configs[2]): 147456 rays, 64 coarse + 64 fine samples, 7 dynamic + 11 static views, chunk 8192; '
                  + ('rays tiled over %d ranks, one packed [rays,5] all-gather: strong scaling' % world if world > 1 else 'one GPU'),
          'n_gpus': world, 'ms_per_frame': fdt * 1e3, 'ms_per_frame_each': [round(t * 1e3, 2) for t in frame_times], 'rays_per_s': n_frame_rays / fdt, 'gather': render_image.GATHER if multi_rank else None,
          'chunk_streams': getattr(render_image, 'CHUNK_STREAMS', 1),
          'per_rank': {'tile_rays': [int(v) for v in every_rank(fst.get('tile_rays', 0))],
                       'render_ms': [round(v, 3) for v in every_rank(fst.get('render_ms', 0.0))],
                       'gather_and_copy_ms': [round(v, 3) for v in every_rank(fst.get('gather_ms', 0.0))],
                       'gather_payload_bytes_per_rank': int(fst.get('gather_bytes', 0)),
                       'chunk_rays_rank0': fst.get('chunk_rays'), 'chunk_ms_rank0': fst.get('chunk_ms'),
                       'note': 'render_ms: the chunk loop over the rank\'s own ray tile; gather_and_copy_ms: the packed [rays,5] all-gather + the copy of the frame\'s pixels '
                               'to the host (render_image.FRAME_STATS: the device is synchronised between the two stages for this frame only)'},
          'kernel_ms_per_frame_rank0_one_stream': {k: round(v['avg_ms'] * v['launches'], 3) for k, v in sorted(fk.items(), key=lambda kv: -kv[1]['avg_ms'] * kv[1]['launches'])},
          'gather_algorithmic_bytes_rank0': (frame_gather or {}).get('bytes') if lib is not None else None,
          'gather_calls_rank0': (frame_gather or {}).get('calls') if lib is not None else None,
          'pixels_check': [float(ret['outputs_fine_ref']['rgb'].mean()), float(ret['outputs_fine_ref']['depth'].mean())]}
      if dry:
        extra['frame_nvi_288x512']['what'] = 'DRY RUN: %d x %d stub rays through the real render_single_image_nvi on %d gloo rank(s)' % (fc.H, fc.W, world)
      del fc, smp, rb, ret
    except Exception as e:
      extra['frame_nvi_288x512'] = {'error': str(e)[:300]}
    if world == 1 and not dry and not a.frame_only:
      try:
        wl11 = StaticStep(dev, R, S, 11, rank)
        _, dt11, k11 = timed(lib, wl11.step, max(5, a.steps // 2), 2, fence)
        n11 = max(5, a.steps // 2)
        extra['views_11'] = {'what': 'the same step at the 11 static source views of the Nvidia eval (eval_nvidia.py:92-119)', 'value': R * n11 / dt11, 'unit': 'rays/s',
                             'ms_per_step': dt11 / n11 * 1e3, 'kernels_avg_ms': {k: round(v['avg_ms'], 5) for k, v in k11.items()},
                             'k_static_views_vs_8_views': k11['k_static_views']['avg_ms'] / kernels['k_static_views']['avg_ms'], 'rows_vs_8_views': 11 / 8}
        del wl11
      except Exception as e:
        extra['views_11'] = {'error': str(e)[:300]}
      try:
        # K1 alone at the dynamic branch's 7 views, inside the same alternation with the network kernels (the V = 7 tile of the frame)
        wl7 = StaticStep(dev, R, S, 7, rank)
        _, dt7, k7 = timed(lib, wl7.step, 5, 2, fence)
        extra['views_7'] = {'what': 'the same step at 7 source views (the dynamic branch\'s count): priced for its k_project_gather only',
                            'ms_per_step': dt7 / 5 * 1e3, 'kernels_avg_ms': {k: round(v['avg_ms'], 5) for k, v in k7.items()}}
        del wl7
      except Exception as e:
        extra['views_7'] = {'error': str(e)[:300]}
      try:
        # the package power and the shader clock WHILE the step runs back to back (rocm-smi samples beside a 4 s loop of the same step): the 2500 TFLOP/s
        # peak the roofline is priced against assumes 2.4 GHz, the part clocks down to its power cap under these kernels
        extra['power_under_step_loop'] = power_sample(wl.step, sync)
      except Exception as e:
        extra['power_under_step_loop'] = {'error': str(e)[:300]}

    if world == 1 and not dry and not a.frame_only:
      try:
        # section 8(f)1: the feature encoder on the 18 source images of one Balloon1 target view (7 dynamic + 11 static, eval_nvidia.py:335-358)
        from dynibar_amd import feature_network, synthetic as syn
        n_img = 18
        imgs = torch.rand(n_img, H, W, 3, device=dev)
        ew = syn.make_encoder_weights(0)
        enc = feature_network.ResNet.from_module(ew)
        x = imgs.permute(0, 3, 1, 2)
        enc(x); fence()
        lib.dyn_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(5):
          xc, xf = enc(x)
        fence()
        edt = (time.perf_counter() - t0) / 5
        lib.dyn_profile_enable(0)
        ek = read_kernels(lib)
        h1, w1, h2, w2 = H // 2, W // 2, H // 4, W // 4
        eflops = n_img * 2.0 * (h1 * w1 * 64 * 147 + h2 * w2 * 64 * (6 * 576 + 2 * 64))
        extra['feature_encoder'] = {'what': 'ResNet feature encoder (feature_network.py:179-311) on 18 source images 288x512 -> 2 x [18,32,72,128], channels-last, HIP implicit-GEMM '
                                            'convolutions on the split-product MFMA engine', 'ms': edt * 1e3, 'algorithmic_tflops': eflops / edt / 1e12,
                                    'frac_of_split_mfma_peak': eflops / edt / 1e12 / split_peak(int(lib.dyn_mlp_split_terms())),
                                    'kernel_ms': {k: round(v['avg_ms'] * v['launches'] / 5, 4) for k, v in ek.items()}}
        try:  # the reference's own route on this GPU: the same convolutions as PyTorch eager ops (MIOpen), via the oracle's restatement
          from oracle import ibr_oracle as O
          sd_dev = {k: torch.from_numpy(v).to(dev) for k, v in ew.items()}
          with torch.no_grad():
            rc, rf = O.resnet_encoder(sd_dev, x.contiguous()); fence()
            t0 = time.perf_counter()
            for _ in range(3):
              rc, rf = O.resnet_encoder(sd_dev, x.contiguous())
            fence()
          extra['feature_encoder']['pytorch_eager_same_gpu_ms'] = (time.perf_counter() - t0) / 3 * 1e3
          extra['feature_encoder']['max_abs_diff_vs_pytorch_eager'] = float((xc - rc).abs().max())
        except Exception as e:
          extra['feature_encoder']['pytorch_eager_same_gpu_ms'] = 'failed: ' + str(e)[:120]
        try:  # training form (forward with saved activations + backward: train.py:272-281 optimises feature_net) and the same through PyTorch eager
          from dynibar_amd import train_encoder
          pt = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in ew.items() if k in train_encoder.PARAMS}
          cot = torch.randn(n_img, 32, H // 4, W // 4, device=dev)

          def train_pass(fn):
            for q in pt.values():
              q.grad = None
            c, f = fn()
            ((c * cot).sum() + (f * cot).sum()).backward()

          ours = lambda: train_encoder.encoder_forward(pt, x)
          train_pass(ours); fence()
          t0 = time.perf_counter()
          for _ in range(3):
            train_pass(ours)
          fence()
          tdt = (time.perf_counter() - t0) / 3
          extra['feature_encoder']['training_form'] = {'what': 'forward with saved activations + backward to all 26 parameter tensors (im2col + dyn_train_gemm + InstanceNorm row kernels), 18 images',
                                                       'ms_forward_backward': tdt * 1e3}
          eager = lambda: O.resnet_encoder(pt, x.contiguous())
          train_pass(eager); fence()
          t0 = time.perf_counter()
          for _ in range(3):
            train_pass(eager)
          fence()
          extra['feature_encoder']['training_form']['pytorch_eager_same_gpu_ms'] = (time.perf_counter() - t0) / 3 * 1e3
        except Exception as e:
          extra['feature_encoder']['training_form'] = {'error': str(e)[:200]}
        del imgs, enc, x
      except Exception as e:
        extra['feature_encoder'] = {'error': str(e)[:300]}

      try:
        # BASELINE configs[3] (kid-running monocular full frame, dual branch, scene-flow-warped views) and configs[4] (stress: 16 + 16 views,
        # 128 + 128 samples, one 8192-ray chunk) as timed legs; their parity is tests/test_gpu_parity.py (mono_kid / stress goldens)
        import config_cases
        with torch.no_grad():
          mf = config_cases.MonoFrame(dev)
          mf.render(); fence()
          t0 = time.perf_counter()
          mf.render()
          fence()
          mdt = time.perf_counter() - t0
          extra['mono_frame_kid'] = {'what': 'ONE render_single_image_mono call (BASELINE configs[3]): 147456 rays, 64 samples, 7 + 3 dynamic and 15 static views, '
                                             'anti_alias_pooling 0 / mask_rgb 1, chunk 8192; one GPU', 'ms_per_frame': mdt * 1e3, 'rays_per_s': mf.rays / mdt}
          del mf
          torch.cuda.empty_cache()
          stc = config_cases.StressChunk(dev)
          stc.render(); fence()
          t0 = time.perf_counter()
          for _ in range(3):
            stc.render()
          fence()
          sdt = (time.perf_counter() - t0) / 3
          extra['stress_chunk'] = {'what': 'ONE render_rays_mv chunk (BASELINE configs[4]): 8192 rays, 128 coarse + 128 fine samples, 16 dynamic + 16 static views; one GPU',
                                   'ms_per_chunk': sdt * 1e3, 'rays_per_s': stc.R / sdt}
          del stc
          torch.cuda.empty_cache()
      except Exception as e:
        extra['config_legs'] = {'error': str(e)[:300]}
      try:
        # section 8(f)2: the per-view body of the reference's evaluation loop (sampler, four encoder passes, full-frame render, pixels to the
        # host, PSNR) -- what the README's "hours per scene" is made of
        import eval_loop
        extra['eval_loop'] = eval_loop.run(dev, views=2)
        torch.cuda.empty_cache()
      except Exception as e:
        extra['eval_loop'] = {'error': str(e)[:300]}
      try:
        # section 8(f)3, first slice: one static bootstrap training step (train.py:116-199) at the reference's training shape
        # (configs/train_kid-running.txt: N_rand 3072, 64 samples, 15 static views, anti_alias_pooling 0, mask_rgb 1): forward with saved
        # activations + backward through the dyn_train_* kernels into DynibarStatic's parameters and the static feature maps
        from dynibar_amd import ops as _ops, synthetic as syn, train_motion as TM, train_static as TS
        Rt, St, Vt = 3072, 64, 15
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()  # peak_mem_gb below is this leg's own peak, not the process's
        sc = syn.make_scene(seed=21, H=H, W=W, V=7, n_static=Vt, smooth=False)
        td = lambda x: torch.from_numpy(x).to(dev)
        fm = td(sc['static_featmaps']).requires_grad_(True)
        tviews = _ops.SourceViews(td(sc['camera']), td(sc['static_src_rgbs']), td(sc['static_src_cameras']), fm.detach())
        to_, td_, _ = syn.pixel_rays(sc['camera'], syn.sample_pixels(21, H, W, Rt))
        to_, td_ = td(to_), td(td_)
        prm = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in syn.make_weights('static', 0).items() if k != 's'}
        tdr = td(sc['depth_range'])
        cot = torch.randn(Rt, 3, device=dev)

        def train_step():
          pts_, z_, _s = _ops.sample_along_ray(to_, td_, tdr, St, True)
          rf_, rd_, mk_, pm_ = TM.gather(tviews, fm, Rt, St, ray_o=to_, ray_d=td_, z_vals=z_, pix_mask_thresh=1.0)
          raw_ = TS.static_raw(prm, (False, True), tviews, rf_, to_, td_, pts_, rd_, mk_)
          (TS.composite_vanilla(raw_, z_, pm_)['rgb'] * cot).sum().backward()

        train_step(); train_step(); fence()
        lib.dyn_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(3):
          train_step()
        fence()
        tdt = (time.perf_counter() - t0) / 3
        lib.dyn_profile_enable(0)
        tk = read_kernels(lib)
        tflop = 3.0 * (0.361e6 + 0.033e6 * (St / 64) + 0.4305e6 * Vt) * Rt * St  # forward + data gradient + weight gradient
        extra['train_static_step'] = {
            'what': 'ONE static bootstrap training step (train.py:116-199): gather -> DynibarStatic -> raw2outputs_vanilla, loss.backward() into the 38 '
                    'parameters and the static feature maps; 3072 rays x 64 samples x 15 views (configs/train_kid-running.txt)',
            'ms_per_step': tdt * 1e3, 'rays_per_s': Rt / tdt, 'algorithmic_tflop_per_step': tflop / 1e12, 'algorithmic_tflops': tflop / tdt / 1e12,
            'frac_of_split3_mfma_peak': tflop / tdt / 1e12 / (2500.0 / 3), 'peak_mem_gb': torch.cuda.max_memory_allocated() / 2**30,
            'kernel_ms': {k: round(v['avg_ms'] * v['launches'] / 3, 3) for k, v in tk.items()}}
        try:  # the reference's own route on this GPU: the same graph as PyTorch eager ops + autograd (the oracle's restatement on the device)
          from oracle import ibr_oracle as O
          sdv = {k: v.detach().clone().requires_grad_(True) for k, v in prm.items()}
          osc = {k: td(sc[k]) for k in ('camera', 'static_src_rgbs', 'static_src_cameras', 'depth_range')}
          ofm = fm.detach().clone().requires_grad_(True)
          osc['static_featmaps'] = ofm

          def eager_step():
            out = O.static_branch_pass(sdv, osc, to_, td_, St, True, True, False, True)
            (out['rgb'] * cot).sum().backward()

          eager_step(); fence()
          t0 = time.perf_counter()
          for _ in range(2):
            eager_step()
          fence()
          extra['train_static_step']['pytorch_eager_same_gpu_ms'] = (time.perf_counter() - t0) / 2 * 1e3
          gk = 'base_fc.2.weight'
          g_ours, g_eager = (prm[gk].grad / 5).double(), (sdv[gk].grad / 3).double()
          extra['train_static_step']['grad_check_vs_pytorch_eager'] = {'tensor': gk, 'max_abs_ours': float(g_ours.abs().max()), 'max_abs_eager': float(g_eager.abs().max()),
                                                                       'max_abs_diff': float((g_ours - g_eager).abs().max())}
        except Exception as e:
          extra['train_static_step']['pytorch_eager_same_gpu_ms'] = 'failed: ' + str(e)[:160]
        del prm, fm, tviews
        torch.cuda.empty_cache()
      except Exception as e:
        extra['train_static_step'] = {'error': str(e)[:300]}

      try:
        # section 8(f)3 complete: the reference's whole main-loop iteration (train.py:203-467) at the kid-running training shape:
        # render_rays_mono(is_train=True) under grad mode + backward into the 3 nets, the trajectory basis and the 3 feature-map sets
        from train_case import TrainCase
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()  # (this leg's own peak)
        leg = {}
        for Rt in (3072, 1024):
          tc = TrainCase(dev, R=Rt)
          tc.step(); tc.step(); fence()
          lib.dyn_profile_enable(1)
          from dynibar_amd import train_static as _ts
          _ts.GEMM_STATS = {'bytes': 0, 'flops': 0, 'calls': 0}
          t0 = time.perf_counter()
          for _ in range(3):
            tc.step()
          fence()
          tdt = (time.perf_counter() - t0) / 3
          lib.dyn_profile_enable(0)
          gst, _ts.GEMM_STATS = _ts.GEMM_STATS, None
          tk = read_kernels(lib)
          fl = tc.algorithmic_flops()
          leg[f'rays_{Rt}'] = {'ms_per_step': tdt * 1e3, 'rays_per_s': Rt / tdt, 'algorithmic_tflop_per_step': fl / 1e12, 'algorithmic_tflops': fl / tdt / 1e12,
                               'frac_of_split3_mfma_peak': fl / tdt / 1e12 / (2500.0 / 3),
                               'kernel_ms': {k: round(v['avg_ms'] * v['launches'] / 3, 3) for k, v in tk.items()}}
          if 'k_train_gemm' in tk:
            gms = tk['k_train_gemm']['avg_ms'] * tk['k_train_gemm']['launches'] / 3
            leg[f'rays_{Rt}']['roofline_train_gemm'] = {
                'kernel': 'k_train_gemm (all %d launches of an iteration: forward, data gradient, weight gradient)' % (gst['calls'] // 3), 'bound': 'hbm',
                'achieved': gst['bytes'] / 3 / gms / 1e6, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gst['bytes'] / 3 / gms / 1e6 / 8000.0,
                'algorithmic_bytes_per_iteration': gst['bytes'] / 3, 'ms_per_iteration': gms, 'algorithmic_tflops': gst['flops'] / 3 / gms / 1e9,
                'frac_of_split3_mfma_peak': gst['flops'] / 3 / gms / 1e9 / (2500.0 / 3),
                'note': 'activations round-trip through HBM between layers (about 40 FLOP per byte): HBM-bound by design; launch time from HIP events on the launch stream'}
          if Rt == 3072:
            try:  # the part of the reference's iteration in front of render_rays_mono: both feature nets on the source views, trained (train.py:264-281)
              from dynibar_amd import synthetic as _syn, train_encoder
              encs = [{k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in _syn.make_encoder_weights(sd).items() if k in train_encoder.PARAMS} for sd in (0, 1)]
              cb = torch.cat([tc.batch['src_rgbs'].squeeze(0), tc.batch['anchor_src_rgbs'].squeeze(0)], 0).permute(0, 3, 1, 2)
              stx = tc.batch['static_src_rgbs'].squeeze(0).permute(0, 3, 1, 2)

              def enc_step():
                for e_ in encs:
                  for q_ in e_.values():
                    q_.grad = None
                a_, _ = train_encoder.encoder_forward(encs[0], cb)
                b_, _ = train_encoder.encoder_forward(encs[1], stx)
                (a_.square().mean() + b_.square().mean()).backward()

              enc_step(); fence()
              t0 = time.perf_counter()
              for _ in range(3):
                enc_step()
              fence()
              leg['rays_3072']['feature_nets_ms'] = (time.perf_counter() - t0) / 3 * 1e3
              leg['rays_3072']['feature_nets_what'] = ('feature_net on %d + feature_net_st on %d source images %dx%d, training form: forward + backward to their parameters '
                                                       '(train.py:264-281; in front of the iteration timed above, which takes the maps as given)' %
                                                       (cb.shape[0], stx.shape[0], cb.shape[2], cb.shape[3]))
              del encs, cb, stx
            except Exception as e:
              leg['rays_3072']['feature_nets_ms'] = 'failed: ' + str(e)[:160]
          if Rt == 1024:
            try:  # the reference's own route on this GPU: the same iteration as PyTorch eager ops + autograd (the oracle's restatement on the device)
              from oracle import ibr_oracle as O
              Wd = {k: {n: v.detach().clone().requires_grad_(True) for n, v in getattr(tc.model, k).items()} for k in ('net_coarse_st', 'net_coarse_dy', 'motion_mlp')}
              Wd['trajectory_basis'] = tc.model.trajectory_basis.detach().clone().requires_grad_(True)
              osc = {k: tc.batch[k] for k in ('camera', 'depth_range', 'src_rgbs', 'src_cameras', 'static_src_rgbs', 'static_src_cameras', 'anchor_src_rgbs',
                                             'anchor_src_cameras')}
              osc['featmaps'], osc['featmaps_anchor'], osc['static_featmaps'] = (f.detach().clone().requires_grad_(True) for f in tc.feat)

              def eager_step():
                ret = O.render_rays_mono_train(Wd, osc, tc.batch['ray_o'], tc.batch['ray_d'], tc.batch['uv_grid'], tc.fidx, tc.temb, tc.toff, tc.S, True, True,
                                               anti_alias_pooling=False, mask_rgb=True, num_vv=tc.num_vv)
                (ret['outputs_coarse_ref']['rgb'] * tc.c_rgb).sum().backward()

              eager_step(); fence()
              t0 = time.perf_counter()
              for _ in range(2):
                eager_step()
              fence()
              leg['rays_1024']['pytorch_eager_same_gpu_ms'] = (time.perf_counter() - t0) / 2 * 1e3
            except Exception as e:
              leg['rays_1024']['pytorch_eager_same_gpu_ms'] = 'failed: ' + str(e)[:160]
          del tc
          torch.cuda.empty_cache()
        leg['what'] = ('ONE iteration of the reference main loop (train.py:203-467): render_rays_mono(is_train=True) + loss.backward() through the HIP training '
                       'kernels; 64 samples, 10 + 10 dynamic (reference + anchor frame) and 15 static views (configs/train_kid-running.txt)')
        leg['peak_mem_gb'] = torch.cuda.max_memory_allocated() / 2**30
        extra['train_full_iteration'] = leg
      except Exception as e:
        extra['train_full_iteration'] = {'error': str(e)[:300]}

  if rank != 0:
    if multi_rank:
      dist.destroy_process_group()
    return

  value = world * R * a.steps / dt
  if dry:
    print(json.dumps({'metric': 'rays/sec (64 samples x 8 src views)', 'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                      'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none (dry run)',
                      'data': 'synthetic', 'dry_run': True,
                      'config': {'workload': 'DRY RUN of the control flow on CPU ranks (gloo) over a stub of the kernel layer: not a measurement',
                                 'rays_per_step_per_gpu': R, 'samples': S, 'src_views': V},
                      'roofline': {'kernel': 'k_static_views', 'bound': 'mfma', 'achieved': None, 'peak': split_peak(3), 'unit': 'TFLOP/s', 'frac': None, 'traffic': None},
                      'multi_gpu': multi, 'extra': extra}))
    if multi_rank:
      dist.destroy_process_group()
    return
  terms, kind = int(lib.dyn_mlp_split_terms()), int(lib.dyn_mlp_split_kind())
  peak = split_peak(terms)
  dom = kernels['k_static_views']
  flops_launch = FLOP_VIEWS_PER_PV * R * S * V
  achieved = flops_launch / (dom['avg_ms'] * 1e-3) / 1e12
  net_ms = sum(kernels[k]['avg_ms'] for k in ('k_static_ref_feat', 'k_static_views', 'k_static_points', 'k_static_blend'))
  net_tflops = static_net_flops_per_point(S, V) * R * S / (net_ms * 1e-3) / 1e12
  pg = kernels['k_project_gather']
  pg_bytes = gather_bytes(R, S, V)
  if terms == 0:
    engine = 'f32'
  elif kind == 2:
    engine = ('f32 in / f32 accumulate; products from two half-float parts per operand (22-bit operands, 3 of the 4 partial products on the f16 matrix pipe, '
              '~2^-20 relative per product); the exact 6-term bf16 build is tested beside it')
  else:
    engine = f'f32 in / f32 accumulate; products from {"three" if terms == 6 else "two"} bf16 parts per operand, {terms} partial products on the bf16 matrix pipe'


  traffic, traffic_note, clock = None, 'skipped (--no-traffic)', None
  if not a.no_traffic and world == 1:
    traffic, traffic_note = pmc_traffic(a)
    ck, _ = pmc_clock(a)
    if ck and ck.get('grbm_gui_active') and ck.get('avg_us_under_profiler'):
      ghz = ck['grbm_gui_active'] / 8.0 / (ck['avg_us_under_profiler'] * 1e3)  # cycles per XCD / ns
      clock = {'effective_shader_clock_ghz': ghz, 'nominal_ghz': 2.4,
               'mfma_pipe_busy_frac': (ck['mfma_busy_cycles'] / (ck['grbm_gui_active'] / 8.0 * 1024.0)) if ck.get('mfma_busy_cycles') else None,
               'note': 'k_static_views under the profiler: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs). '
                       'The 2500 TFLOP/s dense peak assumes 2.4 GHz; frac_of_throttled_peak rescales it to the clock the power budget allowed'}
  tr = lambda k: (traffic[k]['bytes'] if traffic and k in traffic else None)
  insts = None
  if not a.no_traffic and world == 1:
    insts, _ = pmc_insts(a)
  # VALU work per matrix instruction of the dominant kernel -- the lever DESIGN.md section 4 names (the chain sits at its issue bound): tracked per run
  mfma_static = 894 * (R * S * V // 32) if V == 8 else None  # MFMAs per launch from the ISA: 894 per wave of 32 point-views (8-view flavour)
  if clock is not None and insts and 'k_static_views' in insts:
    iv = insts['k_static_views']
    n_valu, n_mfma = iv.get('SQ_INSTS_VALU'), iv.get('SQ_INSTS_MFMA') or mfma_static
    if n_valu and n_mfma:
      non_mfma = n_valu - n_mfma if n_valu > n_mfma else n_valu
      clock.update(valu_insts_per_launch=n_valu, mfma_insts_per_launch=n_mfma, valu_per_mfma=non_mfma / n_mfma, valu_per_product_triple=3.0 * non_mfma / n_mfma,
                   valu_note='SQ_INSTS_VALU (VALU-class instructions issued, matrix instructions included) and SQ_INSTS_MFMA (or, if that counter is '
                             'unavailable, 894 MFMAs per wave from the ISA) per launch; valu_per_product_triple = non-matrix VALU per three partial products')

  # secondary rooflines live INSIDE `roofline` so that a driver which keeps only that object still carries them
  def hbm_obj(kernel, nbytes, ms, traffic_bytes=None, **kw):
    o = {'kernel': kernel, 'bound': 'hbm', 'achieved': nbytes / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
         'frac': nbytes / (ms * 1e-3) / (HBM_PEAK_GBPS * 1e9), 'avg_launch_ms': ms, 'algorithmic_bytes_per_launch': nbytes, 'traffic': traffic_bytes}
    o.update(kw)
    return o

  secondary = {'project_gather_v8': hbm_obj('k_project_gather_tile', pg_bytes, pg['avg_ms'], tr('k_project_gather_tile'),
                                            note='north_star target: >= 0.50 of the HBM roofline on the fused projection / gather kernel')}
  k11 = (extra.get('views_11') or {}).get('kernels_avg_ms') or {}
  if 'k_project_gather' in k11:
    secondary['project_gather_v11'] = hbm_obj('k_project_gather_tile', gather_bytes(R, S, 11), k11['k_project_gather'],
                                              note='the same kernel at the 11 static views of the Nvidia eval (extra.views_11)')
  if 'k_static_blend' in kernels:
    # compulsory bytes of the blend: per point-view the parked x (512), vis (4), the source colour (12), ray_diff (16), mask (4); per point the
    # point part of rgb_fc.0 (512) in, rgb out (12)
    blend_bytes = R * S * V * (512 + 4 + 12 + 16 + 4) + R * S * (512 + 12)
    secondary['static_blend'] = hbm_obj('k_static_blend_ws', blend_bytes, kernels['k_static_blend']['avg_ms'],
                                        tr('k_static_blend_ws') or tr('k_static_blend'),
                                        note='streams the parked per-view feature x back in: HBM-bound; the plain-copy ceiling of the part is ~6.3 TB/s')
  secondary['static_net'] = {'kernels': 'k_static_ref_feat + k_static_views + k_static_points + k_static_blend', 'bound': 'mfma', 'achieved': net_tflops, 'peak': peak,
                             'unit': 'TFLOP/s', 'frac': net_tflops / peak, 'avg_ms': net_ms, 'frac_vs_fp32_mfma_peak': net_tflops / FP32_MFMA_PEAK_TFLOPS}

  # The one-workgroup-per-CU kernels, priced by their own FLOPs (a slow KERNEL shows as a low fraction) and, separately, the signature of the slow STATE some
  # sessions of rounds 4-5 showed (a co-tenant holding registers / LDS: the shader engines have waves resident only ~half of the launch, SQ_BUSY_CYCLES)
  sclk = None
  try:
    sclk = float(np.mean(extra['power_under_step_loop']['shader_clock_mhz'])) / 1e3
  except Exception:
    pass
  state = {'what': 'one-workgroup-per-CU kernels: algorithmic FLOPs / time / split-MFMA ceiling (nominal clock), and the fraction of the launch the shader engines had '
                   'waves resident (SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE / 4: ~0.95 normally, ~0.5 in the slow state of rounds 4-5); slow_state = that fraction < 0.75',
           'shader_clock_ghz': sclk}
  points_frac = motion_frac = dyn_points_frac = None
  if 'k_static_points' in kernels:
    fl = (0.361e6 + 0.033e6 * (S / 64.0)) * R * S
    obs_us = kernels['k_static_points']['avg_ms'] * 1e3
    points_frac = fl / (obs_us * 1e-6) / (peak * 1e12)
    state['k_net_points'] = {'observed_us': obs_us, 'algorithmic_flops_per_launch': fl, 'frac': points_frac, 'vs_k_static_views': obs_us / (dom['avg_ms'] * 1e3)}
  fkm = ((extra.get('frame_nvi_288x512') or {}).get('kernel_ms_per_frame_rank0_one_stream') or {})
  if 'k_motion_mlp' in fkm and world == 1:
    fl = 1.062e6 * 147456 * (58 + 115) * 1.0  # motion MLP FLOPs per frame: the coarse (64) + fine (128) sample points that keep their coefficients
    # (the last round(0.1 S) samples of a ray are zeroed by the reference, render_ray.py:684, and are no longer evaluated: SURVEY 8d's 1.062 MFLOP x 173 of 192 samples)
    motion_frac = fl / (fkm['k_motion_mlp'] * 1e-3) / (peak * 1e12)
    state['k_motion_mlp'] = {'observed_ms_per_frame': fkm['k_motion_mlp'], 'algorithmic_flops_per_frame': fl, 'frac': motion_frac}
  if insts:
    for kn in ('k_net_points', 'k_static_views'):
      iv = insts.get(kn) or {}
      if iv.get('SQ_BUSY_CYCLES') and iv.get('GRBM_GUI_ACTIVE'):
        # (SQ_BUSY_CYCLES is summed over the 32 shader engines, GRBM_GUI_ACTIVE over the 8 XCDs: / 4 = the fraction of the launch the SEs had waves resident)
        state.setdefault('se_busy_fraction', {})[kn] = iv['SQ_BUSY_CYCLES'] / iv['GRBM_GUI_ACTIVE'] / 4.0
  seb = (state.get('se_busy_fraction') or {}).get('k_net_points')
  state['any_slow'] = bool(seb is not None and seb < 0.75)

  # ---- every fraction a reviewer recomputes, as SCALARS directly under `roofline` (a driver that keeps scalars only still carries them) ----
  flat = {}
  flat['k1_v8_us'] = pg['avg_ms'] * 1e3
  flat['k1_v8_frac'] = pg_bytes / (pg['avg_ms'] * 1e-3) / (HBM_PEAK_GBPS * 1e9)
  if 'k_project_gather' in k11:
    flat['k1_v11_us'] = k11['k_project_gather'] * 1e3
    flat['k1_v11_frac'] = gather_bytes(R, S, 11) / (k11['k_project_gather'] * 1e-3) / (HBM_PEAK_GBPS * 1e9)
  k7 = (extra.get('views_7') or {}).get('kernels_avg_ms') or {}
  if 'k_project_gather' in k7:
    flat['k1_v7_us'] = k7['k_project_gather'] * 1e3
    flat['k1_v7_frac'] = gather_bytes(R, S, 7) / (k7['k_project_gather'] * 1e-3) / (HBM_PEAK_GBPS * 1e9)
  fr = extra.get('frame_nvi_288x512') or {}
  if fr.get('gather_algorithmic_bytes_rank0') and 'k_project_gather' in fkm and world == 1:
    flat['k1_inframe_ms'] = fkm['k_project_gather']
    flat['k1_inframe_frac'] = fr['gather_algorithmic_bytes_rank0'] / (fkm['k_project_gather'] * 1e-3) / (HBM_PEAK_GBPS * 1e9)
  if 'static_blend' in secondary:
    flat['blend_us'] = kernels['k_static_blend']['avg_ms'] * 1e3
    flat['blend_frac'] = secondary['static_blend']['frac']
    if secondary['static_blend'].get('traffic'):
      flat['blend_traffic_over_compulsory'] = secondary['static_blend']['traffic'] / secondary['static_blend']['algorithmic_bytes_per_launch']
  if points_frac is not None:
    flat['points_us'] = kernels['k_static_points']['avg_ms'] * 1e3
    flat['points_frac'] = points_frac
  if motion_frac is not None:
    flat['motion_frac'] = motion_frac
  if fr.get('ms_per_frame') and world == 1:
    flat['frame_ms'] = fr['ms_per_frame']
    flat['frame_frac'] = 1.64e9 * 147456 / (fr['ms_per_frame'] * 1e-3) / (peak * 1e12)  # 1.64 GFLOP per ray of the Nvidia eval (64 + 128 samples, 7 + 11 views)
    for kk, nm in (('k_static_views', 'frame_static_views_ms'), ('k_dynamic_views', 'frame_dynamic_views_ms'), ('k_static_points', 'frame_static_points_ms'),
                   ('k_dynamic_points', 'frame_dynamic_points_ms'), ('k_motion_mlp', 'frame_motion_ms'), ('k_static_blend', 'frame_blend_ms'),
                   ('k_trajectory_points', 'frame_trajectory_points_ms')):
      if kk in fkm:
        flat[nm] = fkm[kk]
  if 'views_11' in extra and 'k_static_views_vs_8_views' in extra['views_11']:
    flat['views_11_over_8'] = extra['views_11']['k_static_views_vs_8_views']
  if clock:
    flat['mfma_busy'] = clock.get('mfma_pipe_busy_frac')
    flat['valu_per_mfma'] = clock.get('valu_per_mfma')
    flat['effective_clock_ghz_under_profiler'] = clock.get('effective_shader_clock_ghz')
  flat['shader_clock_ghz'] = sclk
  flat['any_slow'] = state['any_slow']
  flat['step_frac'] = static_net_flops_per_point(S, V) * R * S / (dt / a.steps) / (peak * 1e12)  # one rank's whole step (SURVEY 8d FLOPs of DynibarStatic) against the split ceiling

  res = {
      'metric': 'rays/sec (64 samples x 8 src views)', 'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': a.steps,
      'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': engine, 'data': 'synthetic',
      'config': {'workload': 'BASELINE configs[1]: Nvidia Balloon1 eval shape, static branch only '
                             '(sample -> project/gather -> DynibarStatic -> composite)',
                 'rays_per_step_per_gpu': R, 'samples': S, 'src_views': V, 'src_image': [H, W], 'feature_map': [F, H // 4, W // 4],
                 'sharding': 'ray tiles per rank + RCCL all-gather of rendered pixels' if world > 1 else 'single GPU'},
      'roofline': {'kernel': 'k_static_views', 'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                   'frac': achieved / peak, 'traffic': tr('k_static_views'), 'traffic_note': traffic_note, 'avg_launch_ms': dom['avg_ms'],
                   'algorithmic_flops_per_launch': flops_launch,
                   'peak_note': f'fp32 operands split into {KIND[kind]} parts (two per operand in the shipped engine: 22 mantissa bits, activations by truncation), {terms} '
                                f'partial products per product on the 16-bit matrix pipe, fp32 accumulation: ceiling for ALGORITHMIC fp32 FLOPs = 2500 TFLOP/s dense / {terms}; '
                                'against the native fp32 MFMA peak (157.3 TFLOP/s) see frac_vs_fp32_mfma_peak',
                   'frac_vs_fp32_mfma_peak': achieved / FP32_MFMA_PEAK_TFLOPS, 'clock': clock,
                   'frac_of_throttled_peak': (achieved / (peak * clock['effective_shader_clock_ghz'] / 2.4)) if clock else None,
                   'frac_of_throttled_peak_note': 'NOT a roofline fraction: `frac` with the peak rescaled to the clock the power cap allowed under the profiler; says how much of '
                                                  'the distance to the nominal-clock peak is clock and how much is the kernel',
                   **flat,
                   'secondary': secondary, 'state': state,
                   'kernels_avg_ms': {k: round(v['avg_ms'], 5) for k, v in kernels.items()}},
      'cpu_baseline': None,  # (filled below; kept in front of the long `extra` object)
      'check_vs_oracle': None,
      'multi_gpu': multi,
      'roofline_static_net': secondary['static_net'],
      'roofline_project_gather': secondary['project_gather_v8'],
      'kernels_avg_ms': {k: round(v['avg_ms'], 5) for k, v in kernels.items()},
      'traffic_per_launch': traffic,
      'extra': extra,
  }
  if multi is not None and world == 1:
    multi['note'] = ('ONE rank: the collective path (RCCL all-gather per step, tiled frame) ran with a process group of one; two or more ranks on RCCL have '
                     'only run under gloo on CPU (tests/test_distributed_cpu.py) -- the id broadcast, unequal tiles and deferred-entry collectives meet RCCL '
                     'with N > 1 for the first time on the driver\'s node')

  if a.cpu_rays > 0 and world == 1:
    # the oracle (test infrastructure) is used here ONLY as the timed CPU baseline and as the checker of this run's pixels
    from oracle import ibr_oracle as O
    n = min(a.cpu_rays, R)
    cpu_scene = {k: torch.from_numpy(v) for k, v in wl.sc.items()}
    sd = O.tdict(wl.weights)
    co, cd = torch.from_numpy(wl.o_np[:n]), torch.from_numpy(wl.d_np[:n])
    # torch-CPU oversubscribes on these small per-ray tensors: pick the best thread count on a 32-ray probe, then time the sample 3 times
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
      runs = []
      for _ in range(3):
        t1 = time.perf_counter()
        ref = O.static_branch_pass(sd, cpu_scene, co, cd, S, True, True)
        runs.append(time.perf_counter() - t1)
    cpu_dt = float(np.median(runs))
    res['cpu_baseline'] = {'value': n / cpu_dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
                           'kind_note': 'the oracle (oracle/ibr_oracle.py: the reference algorithm restated on torch-CPU, pinned bit-for-bit to the real reference by '
                                        'tests/golden) -- /root/reference does not exist on the GPU box, so the reference\'s own modules cannot be timed here',
                           'sample': f'{n} of the {R} rays of one step, same scene/weights, torch-CPU oracle, {cores} threads (best of 8/32/64/all on a 32-ray probe; '
                                     f'host has {os.cpu_count()} hardware threads), median of 3 runs after warm-up ({", ".join("%.2f s" % r for r in runs)})'}
    err = (out['rgb'][:n].cpu() - ref['rgb']).abs()
    mse = float((err ** 2).mean())
    res['check_vs_oracle'] = {'rays': n, 'max_abs_rgb_err': float(err.max()), 'psnr_db': (10 * np.log10(1.0 / mse)) if mse > 0 else float('inf')}
  x6 = os.path.join(ROOT, 'dynibar_amd', 'csrc', 'libdynibar_hip_x6.so')
  if world == 1 and a.x6 and os.path.exists(x6) and not os.environ.get('DYNIBAR_HIP_LIB'):
    # the SAME step on the exact engine (three bf16 parts per operand, all 6 partial products down to 2^-18: fp32-class products; ceiling 2500 / 6 TFLOP/s):
    # what exact fp32-class arithmetic costs on this path, and its own distance from the oracle on the same rays
    import tempfile
    dump = os.path.join(tempfile.gettempdir(), 'dynibar_x6_rgb_%d.npy' % os.getpid())
    try:
      r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--steps', str(max(5, a.steps // 2)), '--warmup', '2',
                          '--rays', str(R), '--samples', str(S), '--views', str(V), '--dump-rgb', dump], env=dict(os.environ, DYNIBAR_HIP_LIB=x6),
                         capture_output=True, text=True, timeout=600)
      d6 = json.loads(r.stdout.strip().splitlines()[-1])
      res['x6_engine'] = {'what': 'the same step on libdynibar_hip_x6.so (three bf16 parts per operand, 6 partial products: exact fp32-class products)',
                          'value': d6['value'], 'unit': 'rays/s', 'ms_per_step': d6['ms_per_step'], 'dtype': d6['dtype'],
                          'k_static_views_ms': d6['kernels_avg_ms']['k_static_views'], 'roofline_frac_of_its_own_peak': d6['roofline']['frac'], 'peak': d6['roofline']['peak']}
      res['roofline']['x6_rays_per_s'] = d6['value']
      res['roofline']['x6_over_shipped_time'] = d6['ms_per_step'] / (dt / a.steps * 1e3)
      if res.get('check_vs_oracle') and os.path.exists(dump):
        rgb6 = torch.from_numpy(np.load(dump))
        n6 = min(rgb6.shape[0], ref['rgb'].shape[0])
        e6 = float((rgb6[:n6] - ref['rgb'][:n6]).abs().max())
        res['x6_engine']['max_abs_rgb_err_vs_oracle'] = e6
        res['roofline']['x6_max_abs_rgb_err'] = e6
        res['roofline']['shipped_max_abs_rgb_err'] = res['check_vs_oracle']['max_abs_rgb_err']
    except Exception as e:  # the extra leg must never cost the main line
      res['x6_engine'] = {'error': str(e)[:200]}
    finally:
      if os.path.exists(dump):
        os.remove(dump)
  if a.dump_rgb:
    np.save(a.dump_rgb, out['rgb'][:max(a.cpu_rays, 1024)].detach().cpu().numpy())
  print(json.dumps(res))
  if multi_rank:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
