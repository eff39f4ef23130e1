import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('DYNIBAR_TRAIN_POISON', '1')  # (dynibar_amd/train_static.py: scratch the kernels must fill starts as NaN under test)
for p in (ROOT, os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')
  config.addinivalue_line('markers', 'emu: runs the HIP kernels under the wave-level CPU emulator (test infra)')


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')


def pytest_terminal_summary(terminalreporter, exitstatus, config):
  """How much of each stated tolerance the run used (tests/parity.py:MARGINS) and the chain-level inverse-CDF index report."""
  try:
    import parity
  except Exception:
    return
  if not parity.MARGINS and not parity.CHAIN_INDEX_REPORT:
    return
  import json
  worst = {}
  for m in parity.MARGINS:
    if m['check'] not in worst or m['used'] > worst[m['check']]['used']:
      worst[m['check']] = m
  rows = sorted(worst.values(), key=lambda m: -m['used'])
  tr = terminalreporter
  tr.write_line('')
  tr.write_line(f'parity margins: {len(parity.MARGINS)} tolerance checks, worst fraction of the limit used = {rows[0]["used"]:.2f}' if rows else 'parity margins: none')
  for m in rows[:12]:
    tr.write_line(f'  {m["used"]:5.2f} of limit  err {m["err_at_worst"]:.2e} / {m["limit_at_worst"]:.2e}  (max err {m["max_err"]:.2e})  {m["check"]}')
  for c in parity.CHAIN_INDEX_REPORT:
    tr.write_line(f'  chain-level inverse-CDF indices [{c["case"]}]: {c["mismatches"]} of {c["samples"]} differ from the real reference'
                  + (f' ({c.get("knot_ties", 0)} knot ties within {c.get("tie_bound", 0):.1e})' if c['mismatches'] else ''))
  out = os.path.join(ROOT, 'gpurun_out')
  try:
    os.makedirs(out, exist_ok=True)
    import torch
    # (a CPU session -- oracle and emulator checks -- keeps its own file: it must not overwrite the GPU suite's record, which profiles/ quotes)
    with open(os.path.join(out, 'parity_margins.json' if torch.cuda.is_available() else 'parity_margins_cpu.json'), 'w') as f:
      json.dump({'margins': rows, 'chain_indices': parity.CHAIN_INDEX_REPORT}, f, indent=1)
  except OSError:
    pass
