"""dynibar_amd.criterion against the reference's own ibrnet/criterion.py + utils.img2charbonier (imported live from /root/reference in the
build container; skipped on boxes without it): same values and same gradients for every helper train.py imports."""
import importlib
import sys
import types

import pytest
import torch

import refimport

pytestmark = pytest.mark.skipif(not refimport.have_reference(), reason='/root/reference not present')


def _reference_criterion():
  refimport.import_reference()
  # utils.py imports cv2 / matplotlib at module level for its visualisation helpers; the loss helpers never touch them
  for name in ('cv2', 'matplotlib', 'matplotlib.cm', 'matplotlib.backends', 'matplotlib.backends.backend_agg', 'matplotlib.figure'):
    if name not in sys.modules:
      try:
        importlib.import_module(name)
      except Exception:
        m = types.ModuleType(name)
        m.cm = m.FigureCanvasAgg = m.Figure = object
        sys.modules[name] = m
  return importlib.import_module('ibrnet.criterion')


def _inputs(seed, R=37, V=3):
  g = torch.Generator().manual_seed(seed)
  r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float32)
  outputs = dict(rgb=r(R, 3).requires_grad_(), mask=(r(R) > 0.2), occ_weight_map=r(R).requires_grad_())
  batch = dict(rgb=r(R, 3))
  return outputs, batch, (r(R) > 0.3).float(), r(V, R, 2).requires_grad_(), r(V, R, 2), (r(V, R, 1) > 0.4).float()


@pytest.mark.parametrize('seed', [0, 1])
def test_loss_helpers_match_the_reference(seed):
  import dynibar_amd.criterion as ours
  ref = _reference_criterion()
  outputs, batch, mm, flow, gt_flow, gt_mask = _inputs(seed)
  cases = [
      ('Criterion', lambda m: m.Criterion()(outputs, batch), [outputs['rgb']]),
      ('Criterion+motion_mask', lambda m: m.Criterion()(outputs, batch, mm), [outputs['rgb']]),
      ('temporal_rgb', lambda m: m.compute_temporal_rgb_loss(outputs, batch, mm), [outputs['rgb'], outputs['occ_weight_map']]),
      ('temporal_rgb no mask', lambda m: m.compute_temporal_rgb_loss(outputs, batch), [outputs['rgb'], outputs['occ_weight_map']]),
      ('rgb', lambda m: m.compute_rgb_loss(outputs['rgb'], batch, outputs['mask'].float()), [outputs['rgb']]),
      ('entropy', lambda m: m.compute_entropy(outputs['occ_weight_map']), [outputs['occ_weight_map']]),
      ('flow', lambda m: m.compute_flow_loss(flow, gt_flow, gt_mask), [flow]),
  ]
  for name, fn, leaves in cases:
    a, b = fn(ours), fn(ref)
    assert torch.equal(a, b), f'{name}: {float(a)} vs {float(b)}'
    ga, gb = torch.autograd.grad(a, leaves), torch.autograd.grad(b, leaves)
    for x, y in zip(ga, gb):
      assert torch.equal(x, y), f'{name}: gradients differ'
