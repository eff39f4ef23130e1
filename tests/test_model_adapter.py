"""The drop-in must accept the reference's own model objects: nn.Modules, DataParallel-wrapped (model.py:134-159, :382-397), for BOTH
argument sets the reference ships -- Nvidia eval (anti_alias_pooling=1, mask_rgb=0: configs_nvidia/eval_balloon1_long.txt:21-22) and
the monocular configs (anti_alias_pooling=0, mask_rgb=1: configs/train_kid-running.txt:41-42, where DynibarStatic has NO `s`
parameter, mlp_network.py:330-331).  Packing is host code, so all of this runs without a GPU."""
import types

import numpy as np
import pytest
import torch

import cases
import refimport
import refmodules
from dynibar_amd import ops, render_ray, synthetic as syn

ARG_SETS = {'nvidia_eval': dict(anti_alias_pooling=1, mask_rgb=0), 'kid_running': dict(anti_alias_pooling=0, mask_rgb=1)}


def _args(name):
  return types.SimpleNamespace(input_dir=True, input_xyz=False, occ_weights_mode=0, **ARG_SETS[name])


def _model(mods, args, wrap):
  """A DynibarMono / DynibarFF shaped holder: nets DataParallel-wrapped like model.py:382-397 when wrap is set."""
  W = cases.model_weights(0)
  w = (lambda m: torch.nn.DataParallel(m)) if wrap else (lambda m: m)
  m = types.SimpleNamespace()
  m.net_coarse_st = w(refmodules.load_numpy_state(mods('static', args), W['net_coarse_st']))
  m.net_coarse_dy = w(refmodules.load_numpy_state(mods('dynamic', args, shift=5.0), W['net_coarse_dy']))
  m.motion_mlp = w(refmodules.load_numpy_state(mods('motion', args), W['motion_mlp']))
  m.trajectory_basis = torch.nn.parameter.Parameter(torch.zeros(cases.NUM_FRAMES, cases.NUM_BASIS)).detach().requires_grad_(True)
  return m


def _standin(kind, args, shift=0.0):
  return refmodules.like_reference(kind, args, num_basis=cases.NUM_BASIS, shift=shift)


def _real(kind, args, shift=0.0):
  NET = refimport.import_reference().mlp_network
  if kind == 'static':
    return NET.DynibarStatic(args, in_feat_ch=32, n_samples=64)
  if kind == 'dynamic':
    return NET.DynibarDynamic(args, in_feat_ch=32, n_samples=64, shift=shift)
  return NET.MotionMLP(num_basis=cases.NUM_BASIS)


def _check_adapter(mods, cfg, wrap):
  args = _args(cfg)
  model = _model(mods, args, wrap)
  W = cases.model_weights(0)
  aa, mr = bool(args.anti_alias_pooling), bool(args.mask_rgb)
  if not aa:
    assert 's' not in render_ray._state_dict(model.net_coarse_st) and 'module.s' not in render_ray._state_dict(model.net_coarse_st)
  st = render_ray._static_net(model, 'net_coarse_st', args, 'cpu')
  assert (st.anti_alias_pooling, st.mask_rgb) == (int(aa), int(mr))
  want = dict(W['net_coarse_st'])
  if not aa:
    want.pop('s')
  assert torch.equal(st.blob, ops.StaticNet(want, 'cpu', aa, mr).blob), 'module and dict pack to different blobs'
  dy = render_ray._dynamic_net(model, 'net_coarse_dy', 'cpu')
  assert dy.shift == 5.0 and torch.equal(dy.blob, ops.DynamicNet(W['net_coarse_dy'], 'cpu').blob)
  mo = render_ray._motion_mlp(model, 'motion_mlp', 'cpu', cases.NUM_BASIS)
  assert torch.equal(mo.blob, ops.MotionMLP(W['motion_mlp'], 'cpu', cases.NUM_BASIS).blob)
  # packed once, re-packed when a parameter changes in place (an optimizer step bumps _version)
  assert render_ray._static_net(model, 'net_coarse_st', args, 'cpu') is st
  with torch.no_grad():
    dict(render_ray._unwrap(model.net_coarse_st).named_parameters())['base_fc.0.weight'].mul_(1.5)
  st2 = render_ray._static_net(model, 'net_coarse_st', args, 'cpu')
  assert st2 is not st and not torch.equal(st2.blob, st.blob)


@pytest.mark.parametrize('cfg', sorted(ARG_SETS))
@pytest.mark.parametrize('wrap', [False, True])
def test_adapter_accepts_modules(cfg, wrap):
  _check_adapter(_standin, cfg, wrap)


@pytest.mark.skipif(not refimport.have_reference(), reason='needs /root/reference (build container only)')
@pytest.mark.parametrize('cfg', sorted(ARG_SETS))
@pytest.mark.parametrize('wrap', [False, True])
def test_adapter_accepts_the_real_reference_modules(cfg, wrap):
  _check_adapter(_real, cfg, wrap)


@pytest.mark.skipif(not refimport.have_reference(), reason='needs /root/reference (build container only)')
@pytest.mark.parametrize('cfg', sorted(ARG_SETS))
def test_standins_mirror_the_real_modules(cfg):
  """tests/refmodules.py is what the GPU box feeds the adapter: its state dicts must be the reference's, key for key."""
  args = _args(cfg)
  for kind in ('static', 'dynamic', 'motion'):
    real, mine = _real(kind, args).state_dict(), _standin(kind, args).state_dict()
    assert list(real.keys()) == list(mine.keys()) or sorted(real.keys()) == sorted(mine.keys()), kind
    for k in real:
      assert tuple(real[k].shape) == tuple(mine[k].shape), (kind, k)
  assert ('s' in _real('static', args).state_dict()) == bool(args.anti_alias_pooling)


def test_missing_s_is_an_error_only_with_anti_alias_pooling():
  W = dict(syn.make_weights('static', 0))
  W.pop('s')
  ops.StaticNet(W, 'cpu', anti_alias_pooling=False, mask_rgb=True)
  with pytest.raises(KeyError, match="no 's'"):
    ops.StaticNet(W, 'cpu', anti_alias_pooling=True)


def test_unsupported_module_variants_are_rejected_by_shape():
  """The C packers index with fixed strides: every tensor is shape-checked first."""
  W = dict(syn.make_weights('static', 0))
  bad = dict(W)
  bad['rgb_fc.0.weight'] = np.zeros((32, 33), np.float32)  # DynibarStatic(input_dir=False), mlp_network.py:396-403
  with pytest.raises(ValueError, match='input_dir=False'):
    ops.StaticNet(bad, 'cpu')
  with pytest.raises(RuntimeError, match='32 feature channels'):
    ops.StaticNet(syn.make_weights('static', 0, F=16), 'cpu')
  bad = dict(W)
  bad['base_fc.0.weight'] = W['base_fc.0.weight'][:, :100]
  with pytest.raises(ValueError, match='base_fc.0.weight'):
    ops.StaticNet(bad, 'cpu')
  bad = dict(W)
  bad.pop('vis_fc.2.bias')
  with pytest.raises(KeyError, match='vis_fc.2.bias'):
    ops.StaticNet(bad, 'cpu')
  with pytest.raises(ValueError, match='coeff_linear'):
    ops.MotionMLP(syn.make_weights('motion', 0, num_basis=4), 'cpu', num_basis=6)
  D = dict(syn.make_weights('dynamic', 0))
  D['ref_pts_fc.0.weight'] = D['ref_pts_fc.0.weight'][:, :128]
  with pytest.raises(ValueError, match='ref_pts_fc.0.weight'):
    ops.DynamicNet(D, 'cpu')


@pytest.mark.skipif(not refimport.have_reference(), reason='needs /root/reference (build container only)')
def test_encoder_packs_the_real_reference_module():
  """dynibar_amd.feature_network.ResNet.from_module takes the reference's ResNet (DataParallel-wrapped like model.py:142-151); the
  decoder layers its state dict also carries are ignored like its forward ignores them."""
  from dynibar_amd import feature_network
  FN = refimport.import_reference().feature_network
  net = FN.ResNet(coarse_out_ch=32, fine_out_ch=32, coarse_only=False)
  sd = {k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(2).items()}
  net.load_state_dict(sd, strict=False)
  enc = feature_network.ResNet.from_module(torch.nn.DataParallel(net))
  a = enc._encoder('cpu').blob
  b = ops.Encoder(syn.make_encoder_weights(2), 'cpu').blob
  assert torch.equal(a, b)
  with pytest.raises(ValueError, match='conv1.weight'):
    bad = dict(syn.make_encoder_weights(2)); bad['conv1.weight'] = bad['conv1.weight'][:, :, :5, :5]
    ops.Encoder(bad, 'cpu')


def _checkpoint_files(tmp_path, modules=None):
  """Files in the reference's two checkpoint formats (model.py:424-441 coarse / monocular, :177-190 fine), from seeded weights."""
  W = cases.model_weights(0)
  T = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
  basis = torch.nn.parameter.Parameter(torch.arange(cases.NUM_FRAMES * cases.NUM_BASIS, dtype=torch.float32).reshape(cases.NUM_FRAMES, cases.NUM_BASIS) * 1e-2)
  if modules is not None:  # state dicts taken from the reference's own modules, de-parallelised like model.py:13-15
    st = lambda kind, sd, **kw: refmodules.load_numpy_state(modules(kind, _args('nvidia_eval'), **kw), sd).state_dict()
    W = {k: {n: t.numpy() for n, t in st('static' if '_st' in k else ('dynamic' if '_dy' in k else 'motion'), v).items()} for k, v in W.items()}
  coarse = {'optimizer': {'state': {}, 'param_groups': []}, 'scheduler': {'last_epoch': 3}, 'net_coarse_st': T(W['net_coarse_st']), 'net_coarse_dy': T(W['net_coarse_dy']),
            'feature_net': T(syn.make_encoder_weights(1)), 'feature_net_st': T(syn.make_encoder_weights(2)), 'motion_mlp': T(W['motion_mlp']), 'traj_basis': basis,
            'global_step': 1234}
  fine = {'optimizer': {}, 'scheduler': {}, 'net_fine_st': T(W['net_fine_st']), 'net_fine_dy': T(W['net_fine_dy']), 'feature_net_fine': T(syn.make_encoder_weights(3)),
          'motion_mlp_fine': T(W['motion_mlp_fine']), 'traj_basis_fine': basis.detach() * 2.0, 'global_step': 99}
  pc, pf = str(tmp_path / 'model_coarse.pth'), str(tmp_path / 'model_fine.pth')
  torch.save(coarse, pc)
  torch.save(fine, pf)
  return pc, pf, W


def test_checkpoint_files_load_into_the_adapter(tmp_path):
  """Section 8(f)4: torch.save dictionaries in the reference's checkpoint formats -> a model the renderer packs."""
  from dynibar_amd import checkpoint
  pc, pf, W = _checkpoint_files(tmp_path)
  m = checkpoint.load_model(pc, pf, device='cpu')
  assert m.global_step == 99 and tuple(m.trajectory_basis.shape) == (cases.NUM_FRAMES, cases.NUM_BASIS) and not m.trajectory_basis.requires_grad
  assert float(m.trajectory_basis_fine[1, 1]) == 2.0 * float(m.trajectory_basis[1, 1])
  args = _args('nvidia_eval')
  for name in ('net_coarse_st', 'net_fine_st'):
    assert torch.equal(render_ray._static_net(m, name, args, 'cpu').blob, ops.StaticNet(W[name], 'cpu', True, False).blob)
  dy = render_ray._dynamic_net(m, 'net_coarse_dy', 'cpu')
  assert dy.shift == 5.0, 'a file with feature_net_st is a monocular checkpoint: DynibarMono builds the dynamic net with shift 5 (model.py:304-309)'
  assert torch.equal(dy.blob, ops.DynamicNet(W['net_coarse_dy'], 'cpu').blob)
  assert torch.equal(render_ray._motion_mlp(m, 'motion_mlp_fine', 'cpu', cases.NUM_BASIS).blob, ops.MotionMLP(W['motion_mlp_fine'], 'cpu', cases.NUM_BASIS).blob)
  for name, seed in (('feature_net', 1), ('feature_net_st', 2), ('feature_net_fine', 3)):
    assert torch.equal(getattr(m, name)._encoder('cpu').blob, ops.Encoder(syn.make_encoder_weights(seed), 'cpu').blob)
  m0 = checkpoint.load_model(torch.load(pc, weights_only=False), device='cpu', dynamic_shift=0.0)  # an already-loaded dictionary, explicit shift
  assert render_ray._dynamic_net(m0, 'net_coarse_dy', 'cpu').shift == 0.0 and not hasattr(m0, 'net_fine_st')
  with pytest.raises(KeyError, match='net_coarse_st'):
    checkpoint.load_model({'net_fine_st': {}}, device='cpu')


@pytest.mark.skipif(not refimport.have_reference(), reason='needs /root/reference (build container only)')
def test_checkpoint_of_the_real_reference_modules(tmp_path):
  """The same through state dicts produced by the reference's own nn.Modules (what de_parallel(net).state_dict() saves)."""
  from dynibar_amd import checkpoint
  pc, pf, W = _checkpoint_files(tmp_path, modules=_real)
  m = checkpoint.load_model(pc, pf, device='cpu')
  assert torch.equal(render_ray._static_net(m, 'net_coarse_st', _args('nvidia_eval'), 'cpu').blob, ops.StaticNet(cases.model_weights(0)['net_coarse_st'], 'cpu', True, False).blob)
  FN = refimport.import_reference().feature_network
  net = FN.ResNet(coarse_out_ch=32, fine_out_ch=32)
  torch.save({'net_coarse_st': m.net_coarse_st, 'feature_net': net.state_dict()}, str(tmp_path / 'x.pth'))  # incl. the decoder layers forward never runs
  m2 = checkpoint.load_model(str(tmp_path / 'x.pth'), device='cpu')
  assert m2.feature_net._encoder('cpu').blob.numel() == ops.Encoder(syn.make_encoder_weights(0), 'cpu').blob.numel()
