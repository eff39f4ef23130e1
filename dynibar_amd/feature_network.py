"""Drop-in for the reference's ``ibrnet/feature_network.py`` encoder: ``ResNet(...)(x) -> (x_coarse, x_fine)`` (reference
feature_network.py:179-311), executed by the HIP convolution kernels of csrc/dyn_encoder.hip.

Only the part of ``ResNet.forward`` the reference executes exists here (conv1 -> bn1 -> relu -> layer1 -> out_conv; the decoder
layers the reference constructs are never run, :302-311).  Under grad mode with trainable parameters the call runs the training
form of the encoder (train_encoder.py) and the maps carry the graph into its parameters; otherwise the forward-only kernels.

Layout: the reference's callers pass ``src_rgbs.squeeze(0).permute(0, 3, 1, 2)`` (eval_nvidia.py:335-358), i.e. an NCHW *view* of
channels-last memory; the kernels read that memory as it is.  The returned ``x_coarse`` / ``x_fine`` are NCHW views ([N,32,Hf,Wf],
what the callers index) of channels-last storage, which is exactly the layout the gather kernel taps: ``Projector`` /
``ops.SourceViews`` recognise it and skip the per-target-view NCHW -> NHWC repack.
"""
from __future__ import annotations

import torch

from . import ops


def _unwrap(net):
  return net.module if hasattr(net, 'module') and not isinstance(net, dict) else net


class ResNet(object):
  """``ResNet.from_module(model.feature_net)`` wraps the reference's (optionally DataParallel-wrapped) module or a state dict; the
  weights are packed once and re-packed when a parameter's version counter changes."""

  def __init__(self, encoder='resnet34', coarse_out_ch=32, fine_out_ch=32, norm_layer=None, coarse_only=False, state_dict=None):
    assert encoder in ['resnet18', 'resnet34'], 'the HIP encoder implements the BasicBlock variants the reference instantiates (model.py:56-66)'
    if coarse_only or coarse_out_ch != 32 or fine_out_ch != 32:
      raise NotImplementedError('the HIP encoder is built for coarse_out_ch = fine_out_ch = 32 (every shipped config)')
    self.coarse_out_ch, self.fine_out_ch = coarse_out_ch, fine_out_ch
    self._source = state_dict
    self._packed = {}

  @classmethod
  def from_module(cls, module_or_state_dict):
    return cls(state_dict=module_or_state_dict)

  def load_state_dict(self, state_dict, strict=True):
    self._source = state_dict
    self._packed = {}

  def eval(self):
    return self

  # ---- the nn.Module surface the reference's own model container touches (model.py:117-131 switch_to_train / switch_to_eval,
  # :177-190 save_model -> state_dict, :85-115 optimizer construction -> parameters): delegated to the wrapped module, so an object of
  # this class can sit where `model.feature_net` sits without AttributeError.
  def train(self, mode=True):
    src = _unwrap(self._source)
    if hasattr(src, 'train'):
      src.train(mode)
    return self

  def state_dict(self, *a, **kw):
    src = _unwrap(self._source)
    if src is None:
      raise RuntimeError('dynibar_amd.feature_network.ResNet has no weights')
    return src.state_dict(*a, **kw) if hasattr(src, 'state_dict') else dict(src)

  def parameters(self, recurse=True):
    src = _unwrap(self._source)
    return src.parameters(recurse) if hasattr(src, 'parameters') else iter(())

  def named_parameters(self, *a, **kw):
    src = _unwrap(self._source)
    return src.named_parameters(*a, **kw) if hasattr(src, 'named_parameters') else iter(())

  def to(self, *a, **kw):
    src = _unwrap(self._source)
    if hasattr(src, 'to') and not isinstance(src, dict):
      src.to(*a, **kw)
    elif isinstance(src, dict):
      self._source = {k: (v.to(*a, **kw) if torch.is_tensor(v) else v) for k, v in src.items()}
    self._packed = {}
    return self

  def cuda(self, device=None):
    return self.to('cuda' if device is None else device)

  def _trains(self):
    """True when this call must carry an autograd graph into the encoder: grad mode on and a source with trainable tensors -- a wrapped module's
    parameters (the reference trains feature_net, train.py:272-281) or a state dict whose tensors require grad (what
    train_dist.trainable_parameters and train_encoder.encoder_forward accept).  Such a call runs the training form (train_encoder.py: saved
    activations, backward kernels); everything else the forward-only kernels."""
    if not torch.is_grad_enabled():
      return False
    src = _unwrap(self._source)
    if hasattr(src, 'parameters'):
      return any(p.requires_grad for p in src.parameters())
    if isinstance(src, dict):
      return any(torch.is_tensor(v) and v.requires_grad for v in src.values())
    return False

  def _state(self):
    src = _unwrap(self._source)
    if src is None:
      raise RuntimeError('dynibar_amd.feature_network.ResNet has no weights: construct it with from_module(...) or load_state_dict(...)')
    if hasattr(src, 'state_dict'):
      ver = tuple((p.data_ptr(), p._version) for p in src.parameters())
      return src.state_dict(), ver
    return src, (id(src),) + tuple((v.data_ptr(), v._version) for v in src.values() if torch.is_tensor(v))

  def _encoder(self, device):
    sd, ver = self._state()
    key = (str(device), ver)
    enc = self._packed.get('enc')
    if enc is None or enc[0] != key:
      enc = (key, ops.Encoder(sd, device))
      self._packed['enc'] = enc
    return enc[1]

  def forward(self, x):
    """x [N,3,H,W] -> (x_coarse [N,32,Hf,Wf], x_fine [N,32,Hf,Wf])."""
    assert x.dim() == 4 and x.shape[1] == 3
    if self._trains():
      from . import train_encoder
      return train_encoder.encoder_forward(_unwrap(self._source), x)
    img = x.permute(0, 2, 3, 1)
    if img.dtype != torch.float32 or not img.is_contiguous():
      img = img.float().contiguous()
    coarse, fine = self._encoder(x.device)(img)
    return coarse.permute(0, 3, 1, 2), fine.permute(0, 3, 1, 2)

  __call__ = forward
