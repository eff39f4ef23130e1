"""Drop-in for the reference's ``ibrnet/criterion.py`` (the loss helpers train.py imports: ``Criterion``, ``compute_temporal_rgb_loss``,
``compute_rgb_loss``, ``compute_flow_loss``, ``compute_entropy``; reference criterion.py:21-85, utils.py:32-39).

These are reductions of [rays, 3] / [views, rays, 2] tensors to one scalar per loss term -- the host side of the training step, kept as the
same few torch operations the reference uses so that ``loss.backward()`` enters the HIP backward kernels (dynibar_amd.train_*) with exactly the
reference's cotangents.  The per-sample work (networks, compositing, gather, motion) is in those kernels, not here.
"""
import torch
import torch.nn as nn

EPSILON = 0.001        # criterion.py:19
TINY_NUMBER = 1e-6     # utils.py:22


def img2charbonier(x, y, mask=None, eps=0.001):
  """utils.py:32-39"""
  if mask is None:
    return torch.mean(torch.sqrt((x - y) ** 2 + eps ** 2))
  return torch.sum(torch.sqrt((x - y) ** 2 + eps ** 2) * mask.unsqueeze(-1)) / (torch.sum(mask) * x.shape[-1] + TINY_NUMBER)


class Criterion(nn.Module):
  """criterion.py:21-40"""

  def forward(self, outputs, ray_batch, motion_mask=None):
    pred_mask = outputs['mask'].float()
    if motion_mask is not None:
      pred_mask = pred_mask * motion_mask.float()
    return img2charbonier(outputs['rgb'], ray_batch['rgb'], pred_mask, EPSILON)


def compute_temporal_rgb_loss(outputs, ray_batch, motion_mask=None):
  """criterion.py:43-56"""
  pred_mask = outputs['mask'].float()
  if motion_mask is not None:
    pred_mask = pred_mask * motion_mask
  final_w = (pred_mask * outputs['occ_weight_map']).unsqueeze(-1).repeat(1, 3)
  return torch.sum(final_w * torch.sqrt((outputs['rgb'] - ray_batch['rgb']) ** 2 + EPSILON ** 2)) / (torch.sum(final_w) + 1e-8)


def compute_rgb_loss(pred_rgb, ray_batch, pred_mask):
  """criterion.py:58-62"""
  return img2charbonier(pred_rgb, ray_batch['rgb'], pred_mask, EPSILON)


def compute_entropy(x):
  """criterion.py:79-80"""
  return -torch.mean(x * torch.log(x + 1e-8))


def compute_flow_loss(render_flow, gt_flow, gt_mask):
  """criterion.py:83-85"""
  gt_mask_rep = gt_mask.repeat(1, 1, 2)
  return torch.sum(torch.abs(render_flow - gt_flow) * gt_mask_rep) / (torch.sum(gt_mask_rep) + 1e-8)
