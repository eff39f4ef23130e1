"""Pin the oracle (oracle/ibr_oracle.py) to golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  Same torch build => bit-exact; a different build may differ by rounding in
MKL/ATen kernels, hence the tiny tolerance.  Index/boolean outputs must be exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from oracle import ibr_oracle as O

TOL = dict(rtol=2e-6, atol=2e-6)


def load(golden_dir, name):
  return dict(np.load(os.path.join(golden_dir, name)))


def close(a, b, **kw):
  a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
  if a.dtype == np.bool_ or b.dtype == np.bool_ or np.issubdtype(b.dtype, np.integer):
    np.testing.assert_array_equal(a, b)
  else:
    np.testing.assert_allclose(a, b, **(kw or TOL))


def check_group(prefix, d, g):
  n = 0
  for k, v in d.items():
    if v is None:
      continue
    if isinstance(v, dict):
      n += check_group(prefix + k + '/', v, g)
    else:
      close(v, g[prefix + k]); n += 1
  return n


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_stages(golden_dir, name):
  g = load(golden_dir, f'stages_{name}.npz')
  scene, o, d, uv, pix = cases.scene_case(name)
  S = 64
  W = {k: O.tdict(v) for k, v in cases.model_weights(0).items()}
  W['trajectory_basis'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  W['trajectory_basis_fine'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  for inv in (True, False):
    pts, z, s = O.sample_along_camera_ray(o, d, scene['depth_range'], S, inv, True)
    close(pts, g[f'sample/inv{int(inv)}/pts']); close(z, g[f'sample/inv{int(inv)}/z']); close(s, g[f'sample/inv{int(inv)}/s'])
  pts, z, s = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  Vs = scene['static_src_rgbs'].shape[1]
  rf, rd, mk = O.compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), scene['camera'], scene['static_src_rgbs'],
                                      scene['static_src_cameras'], scene['static_featmaps'])
  close(rf, g['proj_st/rgb_feat']); close(rd, g['proj_st/ray_diff']); close(mk, g['proj_st/mask'], rtol=0, atol=0)
  refc, srcc = O.ref_plucker(o, d), O.src_plucker(pts, scene['static_src_cameras'])
  close(refc, g['plucker/ref']); close(srcc, g['plucker/src'])
  ray_dir = F.normalize(d, dim=-1)
  for aa, mr in ((1, 0), (0, 1)):
    raw = O.static_net(W['net_coarse_st'], pts, refc, srcc, rf, ray_dir, rd, mk, bool(aa), bool(mr))
    close(raw, g[f'static_net/aa{aa}_mr{mr}/raw'])
  raw_st = O.static_net(W['net_coarse_st'], pts, refc, srcc, rf, ray_dir, rd, mk, True, False)
  pm_st = mk[..., 0].sum(dim=2) > 1
  check_group('vanilla_st/', O.raw2outputs_vanilla(raw_st, z, pm_st), g)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  R = pts.shape[0]
  t_emb = temb[None, None, :].repeat(R, S, 1)
  coeff = O.motion_mlp(W['motion_mlp'], torch.cat([pts, t_emb], -1).float())
  close(coeff, g['motion/coeff_raw'])
  coeff[:, -int(round(S * 0.1)):, :] *= 0.0
  traj = O.trajectory_points(coeff, W['trajectory_basis'], fidx)
  pts_seq = torch.stack([pts + (traj[k] - traj[0]) for k in toff], 0)
  close(pts_seq, g['motion/pts_seq'])
  rf_dy, rd_dy, mk_dy = O.compute_with_motions(pts, pts_seq, scene['camera'], scene['src_rgbs'], scene['src_cameras'], scene['featmaps'])
  close(rf_dy, g['proj_dy/rgb_feat']); close(rd_dy, g['proj_dy/ray_diff']); close(mk_dy, g['proj_dy/mask'], rtol=0, atol=0)
  tdiff = (torch.from_numpy(np.array(toff)) / float(cases.NUM_FRAMES))[None, None, :, None].expand(R, S, -1, -1)
  raw_dy = O.dynamic_net(W['net_coarse_dy'], pts, rf_dy, ray_dir, rd_dy, tdiff, mk_dy, t_emb)
  close(raw_dy, g['dynamic_net/raw'])
  pm_dy = mk_dy[..., 0].sum(dim=2) > 1
  comp = O.raw2outputs(raw_dy, raw_st, z, pm_dy, pm_st)
  check_group('composite/', comp, g)
  close(O.compute_optical_flow(comp['weights'], pts_seq, scene['src_cameras'], uv), g['flow/render_flows'])
  w = comp['weights']
  for inv in (True, False):
    if inv:
      iz = 1.0 / z
      bins = torch.flip(0.5 * (iz[:, 1:] + iz[:, :-1]), dims=[1]); ww = torch.flip(w[:, 1:-1], dims=[1])
    else:
      bins = 0.5 * (z[:, 1:] + z[:, :-1]); ww = w[:, 1:-1]
    close(O.sample_pdf(bins, ww, S, det=True), g[f'pdf/inv{int(inv)}/det'])
    close(O.sample_pdf(bins, ww, S, det=False, u=torch.from_numpy(g[f'pdf/inv{int(inv)}/u'])), g[f'pdf/inv{int(inv)}/rand'])
  sc = dict(scene)
  ret = O.render_rays_mv(W, sc, o, d, uv, fidx, temb, toff, S, S, True, True)
  n = check_group('mv/', {k: v for k, v in ret.items() if isinstance(v, dict)}, g)
  assert n == sum(1 for k in g if k.startswith('mv/')), 'oracle mv output key set differs from the reference'
  ret = O.render_rays_mono_eval(W, sc, o, d, uv, fidx, temb, toff, S, True, True, num_vv=0)
  n = check_group('mono/', ret, g)
  assert n == sum(1 for k in g if k.startswith('mono/')), 'oracle mono output key set differs from the reference'


@pytest.mark.parametrize('name', list(cases.CROSS_AXIS_SAMPLES))
def test_cross_axis_shapes(golden_dir, name):
  """render_ray.py:375 / :392 call torch.cross without dim: with exactly 3 source views, a chunk of exactly 3 rays or 3 samples per ray the
  reference's moments are products over THAT axis (views before rays before samples).  The oracle follows it -- pinned here on the reference's
  own Pluecker functions, DynibarStatic on top of them and the static compositing -- and the moments must really differ from the xyz product
  on these shapes (otherwise the cases would pin nothing)."""
  g = {k[len(name) + 1:]: v for k, v in load(golden_dir, 'cross_axis.npz').items() if k.startswith(name + '/')}
  scene, o, d, uv, pix = cases.scene_case(name)
  S = cases.CROSS_AXIS_SAMPLES[name]
  W = O.tdict(cases.model_weights(0)['net_coarse_st'])
  pts, z, s = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  Vs = scene['static_src_rgbs'].shape[1]
  rf, rd, mk = O.compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), scene['camera'], scene['static_src_rgbs'],
                                      scene['static_src_cameras'], scene['static_featmaps'])
  refc, srcc = O.ref_plucker(o, d), O.src_plucker(pts, scene['static_src_cameras'])
  close(refc, g['plucker/ref']); close(srcc, g['plucker/src'])
  c = scene['static_src_cameras'][0, :, -16:].reshape(-1, 4, 4)[:, :3, 3]
  xyz = torch.linalg.cross(c[None, None].expand(pts.shape[0], S, -1, -1), srcc[..., :3], dim=-1)
  assert (xyz - srcc[..., 3:]).abs().max() > 1e-2, 'the source moments of this shape equal the xyz product: the case pins nothing'
  if o.shape[0] == 3:
    assert (torch.linalg.cross(o, refc[:, :3], dim=-1) - refc[:, 3:]).abs().max() > 1e-3  # (the target camera sits near the origin: small moments)
  ray_dir = F.normalize(d, dim=-1)
  for aa, mr in ((1, 0), (0, 1)):
    raw = O.static_net(W, pts, refc, srcc, rf, ray_dir, rd, mk, bool(aa), bool(mr))
    close(raw, g[f'static_net/aa{aa}_mr{mr}/raw'])
    if aa == 1:
      check_group('vanilla_st/', O.raw2outputs_vanilla(raw, z, mk[..., 0].sum(dim=2) > 1), g)


def test_stress_mv(golden_dir):
  """BASELINE configs[4]: the oracle's render_rays_mv at 16 + 16 views, 128 + 128 samples against the real reference's outputs."""
  g = load(golden_dir, 'stress_mv.npz')
  scene, o, d, uv, pix = cases.scene_case('stress')
  W = {k: O.tdict(v) for k, v in cases.model_weights(0).items()}
  W['trajectory_basis'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  W['trajectory_basis_fine'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  ret = O.render_rays_mv(W, dict(scene), o, d, uv, fidx, temb, toff, 128, 128, True, True)
  n = check_group('mv/', {k: v for k, v in ret.items() if isinstance(v, dict)}, g)
  assert n == sum(1 for k in g if k.startswith('mv/')), 'oracle mv output key set differs from the reference'


@pytest.mark.parametrize('tag,shift,mode', [('adj', 1, 0), ('far', -2, 0), ('mode1', 1, 1)])
def test_mono_train(golden_dir, tag, shift, mode):
  """render_rays_mono(is_train=True) forward values (cross-time rendering at the anchor, render_ray.py:1099-1270)."""
  g = load(golden_dir, 'mono_train.npz')
  scene, o, d, uv, pix = cases.scene_case('small')
  sc, fi, te, to = cases.anchor_case(scene, 2, shift)
  W = {k: O.tdict(v) for k, v in cases.model_weights(0).items()}
  W['trajectory_basis'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  ret = O.render_rays_mono_train(W, sc, o, d, uv, fi, te, to, 64, True, True, num_vv=2, occ_weights_mode=mode)
  n = check_group(f'{tag}/', ret, g)
  assert n == sum(1 for k in g if k.startswith(tag + '/')), 'oracle mono train output key set differs from the reference'


def test_mono_kid_config(golden_dir):
  """The monocular configs' arguments (configs/train_kid-running.txt: anti_alias_pooling=0, mask_rgb=1, num_vv=3; dynamic net shift 5.0,
  model.py:304-309), 7 + 3 dynamic and 15 static views, against the real reference's render_rays_mono."""
  g = load(golden_dir, 'mono_kid.npz')
  scene, o, d, uv, pix = cases.scene_case('kid')
  W = {k: O.tdict(v) for k, v in cases.model_weights(0).items()}
  W['trajectory_basis'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  fidx, temb, toff = cases.time_args(7)
  ret = O.render_rays_mono_eval(W, dict(scene), o, d, uv, fidx, temb, toff, 64, True, True, anti_alias_pooling=False, mask_rgb=True, num_vv=3,
                                dy_shift=5.0)
  n = check_group('mono/', ret, g)
  assert n == len(g), 'oracle mono (kid-running arguments) output key set differs from the reference'


def test_chain_level_indices(golden_dir):
  """The inverse-CDF indices recorded INSIDE the reference's render_rays_mv (from its own coarse weights) are what the oracle derives
  from the golden coarse weights: pins the index restatement the GPU chain test compares against."""
  for name in ('small', 'harsh', 'noise'):
    g = load(golden_dir, f'stages_{name}.npz')
    z = torch.from_numpy(g['mv/outputs_coarse_ref/z_vals'])
    w = torch.from_numpy(g['mv/outputs_coarse_ref/weights'])
    _, inds = O.fine_z_vals(z, w, 64, True, True, None, return_inds=True)
    np.testing.assert_array_equal(inds.numpy(), g['chain/mv_above_inds'])


@pytest.mark.parametrize('name', ['small', 'odd', 'wide'])
def test_encoder(golden_dir, name):
  """Section 8(f)1: the oracle's restatement of the executed part of ResNet.forward against the real reference's feature maps."""
  g = load(golden_dir, 'encoder.npz')
  imgs, sd = cases.encoder_case(name)
  xc, xf = O.resnet_encoder(O.tdict(sd), imgs.permute(0, 3, 1, 2))
  close(xc, g[f'{name}/coarse'], rtol=1e-5, atol=1e-5); close(xf, g[f'{name}/fine'], rtol=1e-5, atol=1e-5)


def test_image_rays(golden_dir):
  g = load(golden_dir, 'sampler.npz')
  scene, *_ = cases.scene_case('small')
  o, d, uv = O.image_rays(scene['camera'])
  close(o, g['rays_o']); close(d, g['rays_d']); close(uv, g['uv_grid'])
  _, d2, _ = O.image_rays(scene['camera'], render_stride=2)
  close(d2, g['stride2/rays_d'])


@pytest.mark.parametrize('kid', [True, False])
def test_static_bootstrap_gradients(golden_dir, kid):
  """Section 8(f)3: torch autograd through the oracle's restatement of the static bootstrap graph (train.py:116-199) against the REAL
  reference's autograd on its own modules: loss, rendered colours and the gradient of every DynibarStatic parameter and of the static
  feature maps.  Pins the gradients the GPU tests compare the HIP backward kernels with."""
  import parity
  g = load(golden_dir, 'train_static.npz')
  c = cases.bootstrap_case(kid)
  name = c['name']
  loss, grads, rgb = parity.oracle_bootstrap_step(kid, torch.from_numpy(g[f'{name}/w']))
  close(rgb, g[f'{name}/rgb'], rtol=1e-5, atol=2e-6)
  close(loss, g[f'{name}/loss'], rtol=1e-5, atol=1e-7)
  assert set(grads) == {k[len(name) + 6:] for k in g if k.startswith(f'{name}/grad/')}, 'gradient key set differs from the reference'
  gmax = max(float(np.abs(g[f'{name}/grad/{k}']).max()) for k in grads if k != 'featmaps')
  for k, v in grads.items():
    ref = g[f'{name}/grad/{k}']
    scale = float(np.abs(ref).max())
    # same ATen operators in the same order: fp32 round-off of a different graph shape only (the pooling temperature `s` is the
    # ill-conditioned one, see parity.check_train_static)
    close(v.reshape(ref.shape), ref, rtol=1e-3 if k != 's' else 5e-2, atol=2e-5 * scale + 1e-7 * gmax)


@pytest.mark.parametrize('lname', ['full', 'flow', 'cycle'])
def test_mono_train_gradients(golden_dir, lname):
  """Section 8(f)3: torch autograd through the oracle's render_rays_mono_train + the restated train.py loss against the REAL reference's
  autograd (digests of every gradient): pins the gradients the GPU tests hold the HIP backward kernels to."""
  import parity
  g = load(golden_dir, 'mono_train_grad.npz')
  loss, grads = parity.oracle_mono_train_step(cases.MONO_TRAIN_LOSSES[lname])
  close(loss, g[f'{lname}/loss'], rtol=1e-5, atol=1e-7)
  n = 0
  gmax = max(float(v[0]) for k, v in g.items() if k.startswith(lname + '/') and k.endswith('/absmax') and '/featmaps' not in k)  # round-off floor of the small tensors
  for k, v in grads.items():
    if f'{lname}/{k}/proj' not in g:
      assert v is None or float(v.abs().max()) == 0.0, f'oracle produces a gradient for {k} that the reference does not'
      continue
    assert v is not None, f'oracle has no gradient for {k}'
    d = cases.grad_digest(v)
    am = float(g[f'{lname}/{k}/absmax'][0])
    close(d['proj'], g[f'{lname}/{k}/proj'], rtol=1e-3, atol=(2e-5 * am + 2e-7 * gmax) * v.numel() ** 0.5 + 1e-12)
    close(d['head'], g[f'{lname}/{k}/head'], rtol=1e-3, atol=2e-5 * am + 2e-7 * gmax + 1e-12)
    n += 1
  assert n > (80 if lname == 'full' else 15)  # the cycle / flow terms reach MotionMLP and the basis (and, for flow, the nets through the weights)
