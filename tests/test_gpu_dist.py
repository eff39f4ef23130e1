"""RCCL on the one GPU there is (section 8e): the multi-rank code paths -- process group on the nccl (= RCCL) backend, the per-step pixel all-gather and
barrier-bracketed fences of bench.py, the ray-tiled frame functions with their packed pixel gather, and the C-ABI's dyn_gather_tiles on its own
communicator -- executed with a process group of ONE rank on an MI355X, launched exactly as the driver launches N ranks (torch.distributed.run).
The scaling curve itself needs the driver's 8-GPU node; this makes sure that node is not the first place the code meets RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _run(args, timeout=900):
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + args
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_frame_functions_and_abi_gather_on_one_rccl_rank():
  r = _run([os.path.join(ROOT, 'tests', 'dist_one_rank.py')])
  assert r.returncode == 0 and 'DIST_ONE_RANK_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
  assert 'gather=torch' in r.stdout and 'gather=abi' in r.stdout


@pytest.mark.parametrize('gather', ['torch', 'abi'])
def test_bench_multi_rank_path_on_one_rccl_rank(gather):
  """bench.py --force-dist: the N > 1 control flow of the bench (nccl process group, all_gather_into_tensor per step, dist.barrier() fences, max over ranks,
  the frame leg tiled through render_single_image_nvi) with one rank."""
  r = _run([os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--gather', gather, '--steps', '4', '--warmup', '1', '--cpu-rays', '0', '--no-traffic',
            '--frame-only'])
  assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1
  d = json.loads(lines[0])
  assert d['n_gpus'] == 1 and d['steps'] == 4 and d['value'] > 0
  assert d['multi_gpu'] is not None and d['multi_gpu']['backend'] == 'nccl' and d['multi_gpu']['allgather_bytes_per_rank_per_step'] == 4096 * 4 * 4
  fr = d['extra']['frame_nvi_288x512']
  assert 'error' not in fr, fr
  assert fr['per_rank']['tile_rays'] == [288 * 512] and fr['per_rank']['gather_payload_bytes_per_rank'] == 288 * 512 * 5 * 4
  assert fr['gather'] == gather
