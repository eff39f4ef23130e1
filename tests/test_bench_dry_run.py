"""bench.py's multi-rank command, rehearsed on CPU: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --dry-run` runs the
N > 1 control flow of the bench (process group, per-step pixel all-gather, barrier-bracketed timing, max over ranks, the ray-tiled frame leg through
the REAL render_single_image_nvi, rank 0's single JSON line) on two gloo ranks over a stub of the kernel layer.  The product has no CPU path, so the
dry run measures nothing; it exists so that the first execution on an 8-GPU node cannot fail for a reason a CPU could have found."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _json_line(stdout):
  lines = [l for l in stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, 'exactly one JSON line (rank 0)'
  return json.loads(lines[0])


def test_single_rank_dry_run():
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run', '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=300, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-2000:]
  d = _json_line(r.stdout)
  assert d['dry_run'] is True and d['n_gpus'] == 1 and d['multi_gpu'] is None and d['steps'] == 2 and d['warmup'] == 1
  assert d['extra']['frame_nvi_288x512']['per_rank']['tile_rays'] == [36 * 64]


def test_two_rank_dry_run_through_torch_distributed_run():
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
         os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run']
  r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-2000:]
  d = _json_line(r.stdout)
  assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
  assert 'cpu_baseline' not in d and 'roofline' in d, 'N > 1 lines carry the roofline object and no CPU baseline'
  m = d['multi_gpu']
  assert len(m['per_rank_ms_per_step']) == 2 and m['allgather_alone_ms_per_step'] > 0 and m['allgather_bytes_per_rank_per_step'] == 4096 * 4 * 4
  # value = the rays ALL ranks rendered / the slowest rank's time
  assert abs(d['value'] - 2 * 4096 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
  assert d['ms_per_step'] >= max(m['per_rank_ms_per_step']) - 1e-3
  f = d['extra']['frame_nvi_288x512']
  assert f['n_gpus'] == 2 and f['per_rank']['tile_rays'] == [1152, 1152] and len(f['per_rank']['render_ms']) == 2 and len(f['per_rank']['gather_and_copy_ms']) == 2
  assert f['per_rank']['gather_payload_bytes_per_rank'] == 1152 * 5 * 4, 'one packed [rays, 5] all-gather per frame'
  # the tiled frame's pixels are the single-process frame's pixels
  r1 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run', '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300, cwd=ROOT)
  assert _json_line(r1.stdout)['extra']['frame_nvi_288x512']['pixels_check'] == f['pixels_check']


def test_wrong_world_size_is_refused():
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True, timeout=120, cwd=ROOT,
                     env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
  assert r.returncode != 0 and 'torch.distributed.run' in (r.stderr + r.stdout)
