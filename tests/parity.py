"""Parity checks HIP-path-vs-oracle, parameterised by device so the same assertions run
(a) on a real MI355X through libdynibar_hip.so (-m gpu) and (b) under the wave-level emulator on CPU tensors (tests/emu).

Tolerances (stated per check): integer / boolean / index outputs exact; depths, points and normalised distances bit-exact
(the geometry TU is built with -ffp-contract=off); interpolated colours / features within 2e-5 on smooth maps (coordinate
rounding times the map gradient); network outputs and rendered colours within 1e-4 (BASELINE.json north_star).  Where the
reference's own arithmetic is ill-conditioned (white-noise maps: a 1e-4 px shift of a tap is visible; anti-alias pooling weights: a
cancellation) the allowance added is MEASURED on the oracle per element (projection_sensitivity, check_static_net's exp jitter), not
a blanket factor.  Every check records the fraction of its limit it used (MARGINS, printed by tests/conftest.py)."""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

import cases
from dynibar_amd import ops
from oracle import ibr_oracle as O


def to_dev(x, device):
  if isinstance(x, dict):
    return {k: to_dev(v, device) for k, v in x.items()}
  return x.to(device) if isinstance(x, torch.Tensor) else x


def cpu(x):
  return x.detach().cpu()


# Every tolerance check records how much of its limit it used: (label, max |err|, limit at that element, worst err / limit).
# tests/conftest.py prints the table at the end of the session (and writes gpurun_out/parity_margins.json), so the GPU test record
# shows the margins, not just "passed".
MARGINS = []


def record_margin(what, err, lim):
  err, lim = torch.as_tensor(err).double().reshape(-1), torch.as_tensor(lim).double().reshape(-1)
  if err.numel() == 0:
    return
  ratio = err / lim.clamp(min=1e-300)
  i = int(torch.argmax(ratio))
  MARGINS.append(dict(check=what, max_err=float(err.max()), err_at_worst=float(err[i]), limit_at_worst=float(lim[i]), used=float(ratio[i])))


def assert_close(a, b, atol, rtol=0.0, what='', extra=None):
  """|a - b| <= atol + rtol |b| (+ extra, a per-element conditioning allowance computed by the caller)."""
  a, b = cpu(a).double(), cpu(b).double()
  err = (a - b).abs()
  lim = atol + rtol * b.abs()
  if extra is not None:
    lim = lim + extra.double()
  record_margin(what, err, lim.expand_as(err))
  if not bool((err <= lim).all()):
    i = int(torch.argmax(err - lim))
    raise AssertionError(f'{what}: max err {float(err.max()):.3e} (limit {atol:.1e}+{rtol:.1e}*|ref|) at flat index {i}: '
                         f'{float(a.flatten()[i])} vs {float(b.flatten()[i])}; {int((err > lim).sum())}/{err.numel()} over')


def assert_bitexact(a, b, what=''):
  a, b = cpu(a), cpu(b)
  if not torch.equal(a, b):
    d = (a.double() - b.double()).abs()
    raise AssertionError(f'{what}: not bit-exact, {int((a != b).sum())}/{a.numel()} differ, max abs {float(d.max()):.3e}')


def check_sampling(device, name='small', S=64):
  scene, o, d, uv, _ = cases.scene_case(name)
  dr = scene['depth_range']
  g = torch.Generator().manual_seed(3)
  t_rand = torch.rand(o.shape[0], S, generator=g)
  for inv in (True, False):
    for tr in (None, t_rand):
      pts_r, z_r, s_r = O.sample_along_camera_ray(o, d, dr, S, inv, tr is None, tr)
      pts, z, s = ops.sample_along_ray(o.to(device), d.to(device), dr.to(device), S, inv, None if tr is None else tr.to(device))
      assert_bitexact(z, z_r, f'z_vals inv={inv} det={tr is None}')
      assert_bitexact(pts, pts_r, f'pts inv={inv} det={tr is None}')
      assert_bitexact(s, s_r, f's_vals inv={inv} det={tr is None}')
  # fine-pass helper
  pts_r, z_r, s_r = O.sample_along_camera_ray(o, d, dr, S, True, True)
  pts, s = ops.points_from_z(o.to(device), d.to(device), z_r.to(device), dr.to(device))
  assert_bitexact(pts, pts_r, 'points_from_z pts')
  assert_bitexact(s, s_r, 'points_from_z s_vals')


def _mask_check(mask, mask_ref, pix_margin, what):
  """Masks must be identical except where the oracle's own pixel location sits within ~1e-3 px of the frustum boundary
  (an fp32 tie: the reference's LU inverse and ours differ in the last bits of K.inv(c2w))."""
  diff = cpu(mask).reshape(-1) != mask_ref.reshape(-1)
  bad = diff & ~pix_margin.reshape(-1)
  assert int(bad.sum()) == 0, f'{what}: {int(bad.sum())} mask mismatches away from the boundary'
  return int(diff.sum())


def boundary_margin(xyz, cams, tol=2e-3):
  """[R,S,V] bool: oracle pixel location within tol px of an inbound() edge or depth within 1e-5 of the z=0 plane."""
  pix, _ = O.compute_projections(xyz, cams)
  h, w = cams[0][:2]
  K = cams[:, 2:18].reshape(-1, 4, 4)
  c2w = cams[:, -16:].reshape(-1, 4, 4)
  xyz_h = torch.cat([xyz.reshape(xyz.shape[0], -1, 3), torch.ones(xyz.shape[0], xyz[0].numel() // 3, 1)], -1)
  zc = K.bmm(torch.inverse(c2w)).bmm(xyz_h.permute(0, 2, 1))[:, 2].reshape(pix.shape[:-1])
  near = ((pix[..., 0].abs() < tol) | ((pix[..., 0] - (w - 1)).abs() < tol) | (pix[..., 1].abs() < tol) |
          ((pix[..., 1] - (h - 1)).abs() < tol) | (zc.abs() < 1e-5))
  return near.permute(1, 2, 0)


def check_ray_diff(rd, rd_ref, xyz_st, xyz, qcam, cams, what):
  """ray_diff = [unit(a-b), a.b]: the direction is ill-conditioned when the two unit rays nearly coincide, so its tolerance
  scales with 1/|a-b| (error model: 4 ulp of a unit vector / |a-b|); the dot product is checked to 1e-6."""
  V = xyz.shape[0]
  a = F.normalize(qcam[-16:].reshape(4, 4)[:3, 3][None, None, None] - xyz_st[None], dim=-1)
  b = F.normalize(cams[:, -16:].reshape(-1, 4, 4)[:, :3, 3][:, None, None] - xyz, dim=-1)
  nrm = (a - b).norm(dim=-1).permute(1, 2, 0).clamp(min=1e-6)
  err = (cpu(rd) - rd_ref).abs()
  assert float(err[..., 3].max()) <= 1e-6, f'{what} dot: {float(err[..., 3].max()):.3e}'
  scaled = float((err[..., :3] * nrm[..., None]).max())
  assert scaled <= 1e-6, f'{what} direction: scaled err {scaled:.3e}'


def check_project_gather(device, name='small', S=64):
  scene, o, d, uv, _ = cases.scene_case(name)
  atol = 3e-5  # + on white-noise maps the measured conditioning of the reference's own taps (projection_sensitivity)
  pts_r, z_r, _ = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  R = o.shape[0]
  sd = to_dev(scene, device)
  # static branch: points generated in-kernel from (o, d, z)
  Vs = scene['static_src_rgbs'].shape[1]
  xyz = pts_r[None].repeat(Vs, 1, 1, 1)
  rf_r, rd_r, mk_r = O.compute_with_motions(pts_r, xyz, scene['camera'], scene['static_src_rgbs'], scene['static_src_cameras'],
                                            scene['static_featmaps'])
  views = ops.SourceViews(sd['camera'], sd['static_src_rgbs'], sd['static_src_cameras'], sd['static_featmaps'])
  rf, rd, mk = ops.project_gather(views, R, S, ray_o=o.to(device), ray_d=d.to(device), z_vals=z_r.to(device))
  margin = boundary_margin(xyz, scene['static_src_cameras'][0])
  nflip = _mask_check(mk, mk_r, margin, f'{name} static mask')
  keep = (~margin)[..., None].float()
  sens = projection_sensitivity(lambda: dict(rf=O.compute_with_motions(pts_r, xyz, scene['camera'], scene['static_src_rgbs'],
                                                                          scene['static_src_cameras'], scene['static_featmaps'])[0]))['rf']
  assert_close(cpu(rf) * keep, rf_r * keep, atol, 1e-5, f'{name} static rgb_feat', extra=SENS_FACTOR * sens * keep)
  check_ray_diff(rd, rd_r, pts_r, xyz, scene['camera'][0], scene['static_src_cameras'][0], f'{name} static ray_diff')
  # dynamic branch: explicit displaced points (Projector API form)
  g = torch.Generator().manual_seed(9)
  V = scene['src_rgbs'].shape[1]
  xyz = pts_r[None] + 0.05 * torch.randn(V, R, S, 3, generator=g)
  rf_r, rd_r, mk_r = O.compute_with_motions(pts_r, xyz, scene['camera'], scene['src_rgbs'], scene['src_cameras'], scene['featmaps'])
  views = ops.SourceViews(sd['camera'], sd['src_rgbs'], sd['src_cameras'], sd['featmaps'])
  rf, rd, mk = ops.project_gather(views, R, S, pts_st=pts_r.to(device), xyz=xyz.to(device))
  margin = boundary_margin(xyz, scene['src_cameras'][0])
  nflip += _mask_check(mk, mk_r, margin, f'{name} dynamic mask')
  keep = (~margin)[..., None].float()
  sens = projection_sensitivity(lambda: dict(rf=O.compute_with_motions(pts_r, xyz, scene['camera'], scene['src_rgbs'], scene['src_cameras'],
                                                                          scene['featmaps'])[0]))['rf']
  assert_close(cpu(rf) * keep, rf_r * keep, atol, 1e-5, f'{name} dynamic rgb_feat', extra=SENS_FACTOR * sens * keep)
  check_ray_diff(rd, rd_r, pts_r, xyz, scene['camera'][0], scene['src_cameras'][0], f'{name} dynamic ray_diff')
  pm = ops.sample_mask(mk, 1.0)
  assert_bitexact(pm, (cpu(mk)[..., 0].sum(dim=2) > 1).float(), 'sample_mask')
  # the same mask as a by-product of the gather launch, both thresholds the renderer uses
  for th in (1.0, 0.0):
    rf2, rd2, mk2, pm2 = ops.project_gather(views, R, S, pts_st=pts_r.to(device), xyz=xyz.to(device), pix_mask_thresh=th)
    assert_bitexact(pm2, (cpu(mk)[..., 0].sum(dim=2) > th).float(), f'gather pix_mask > {th}')
    assert_bitexact(rf2, rf, 'gather is deterministic'); assert_bitexact(mk2, mk, 'gather mask deterministic'); assert_bitexact(rd2, rd, 'ray_diff deterministic')
  return nflip


def reference_proj_matrices(cams):
  """K . inv(c2w) [V,4,4] as the reference forms it (projection.py:42-47: torch.inverse + bmm in fp32), on the CPU like the goldens"""
  cams = cams.reshape(-1, 34).float().cpu()
  return cams[:, 2:18].reshape(-1, 4, 4).bmm(torch.inverse(cams[:, -16:].reshape(-1, 4, 4)))


def check_project_gather_same_matrix(device, name='small', S=64):
  """The gather with the projection matrices handed in (ops.SourceViews(proj_matrices=...), what Projector(matrix_mode='torch') does): the kernel then
  starts from the SAME fp32 K.inv(c2w) as the oracle, and what is left between them is the rounding of a 4-term dot product and IEEE-identical
  divisions.  No conditioning allowance, no ray is dropped: masks are compared on every point-view (flips are counted and must be isolated ties on a
  frustum edge), colours / features wherever both masks agree."""
  scene, o, d, uv, _ = cases.scene_case(name)
  pts_r, z_r, _ = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  R = o.shape[0]
  sd = to_dev(scene, device)
  g = torch.Generator().manual_seed(9)
  total, flips, worst = 0, 0, 0.0
  for branch in ('static', 'dynamic'):
    if branch == 'static':
      rgbs, cams, fm = scene['static_src_rgbs'], scene['static_src_cameras'], scene['static_featmaps']
      xyz = pts_r[None].repeat(rgbs.shape[1], 1, 1, 1)
    else:
      rgbs, cams, fm = scene['src_rgbs'], scene['src_cameras'], scene['featmaps']
      xyz = pts_r[None] + 0.05 * torch.randn(rgbs.shape[1], R, S, 3, generator=g)
    rf_r, rd_r, mk_r = O.compute_with_motions(pts_r, xyz, scene['camera'], rgbs, cams, fm)
    views = ops.SourceViews(sd['camera'], rgbs.to(device), cams.to(device), fm.to(device), proj_matrices=reference_proj_matrices(cams[0]))
    rf, rd, mk = ops.project_gather(views, R, S, pts_st=pts_r.to(device), xyz=xyz.to(device))
    same = (cpu(mk) == mk_r)
    nf = int((~same).sum())
    flips += nf
    total += mk_r.numel()
    if nf:
      # a flip is legitimate only as a tie: the oracle's own pixel sits within an ulp-scale distance of an inbound() edge / the z = 0 plane
      tie = boundary_margin(xyz, cams[0], tol=1e-4)
      assert bool(tie[(~same)[..., 0]].all()), f'{name} {branch}: mask differs away from a frustum edge'
    # measured on the MI355X and under the emulator: IDENTICAL bits (the dot product, the two IEEE divisions and the bilinear blend round alike)
    assert_bitexact(cpu(rf) * same, rf_r * same, f'{name} {branch} rgb_feat from the reference\'s own projection matrices')
    worst = max(worst, float(((cpu(rf) - rf_r).abs() * same).max()))
    check_ray_diff(rd, rd_r, pts_r, xyz, scene['camera'][0], cams[0], f'{name} {branch} ray_diff (same matrices)')
  print(f'  gather with the reference\'s own K.inv(c2w) [{name}]: {flips} of {total} mask bits differ; worst |rgb_feat| difference {worst:.2e}')
  return flips, worst


def check_projector_helpers(device, name='small', S=16):
  """Projector.inbound / normalize / compute_projections / compute_angle (projection.py:13-101) against the oracle's restatements."""
  from dynibar_amd import projection
  scene, o, d, uv, _ = cases.scene_case(name)
  pts, z, _ = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  cams = scene['src_cameras'][0]
  V = cams.shape[0]
  g = torch.Generator().manual_seed(4)
  xyz = pts[None] + 0.05 * torch.randn(V, pts.shape[0], S, 3, generator=g)
  pj = projection.Projector(device, matrix_mode=reference_proj_matrices)
  pix, front = pj.compute_projections(xyz.to(device), cams.to(device))
  pix_r, front_r = O.compute_projections(xyz, cams)
  assert tuple(pix.shape) == tuple(pix_r.shape) and front.dtype == torch.bool
  assert_bitexact(front, front_r, f'{name} compute_projections in-front mask')
  assert_close(pix, pix_r, 0.0, 1e-6, f'{name} compute_projections pixel locations (relative: one rounding of a 4-term dot product)')
  rd = pj.compute_angle(pts[None].expand(V, -1, -1, -1).to(device), xyz.to(device), scene['camera'][0].to(device), cams.to(device))
  rd_r = O.compute_angle(pts[None].expand(V, -1, -1, -1), xyz, scene['camera'][0], cams)
  assert tuple(rd.shape) == tuple(rd_r.shape)
  check_ray_diff(rd.permute(1, 2, 0, 3), rd_r.permute(1, 2, 0, 3), pts, xyz, scene['camera'][0], cams, f'{name} compute_angle')
  h, w = cams[0][:2]
  assert_bitexact(pj.inbound(pix_r.to(device), h, w), (pix_r[..., 0] <= w - 1.0) & (pix_r[..., 0] >= 0) & (pix_r[..., 1] <= h - 1.0) & (pix_r[..., 1] >= 0),
                  f'{name} inbound')
  nrm = pj.normalize(pix.reshape(V, -1, 2), h, w)
  assert_bitexact(nrm, 2 * cpu(pix).reshape(V, -1, 2) / torch.tensor([w - 1.0, h - 1.0])[None, None, :] - 1.0, f'{name} normalize')
  return float((cpu(rd) - rd_r).abs().max())


def check_composite(device, R=37, S=64, seed=0):
  g = torch.Generator().manual_seed(seed)
  raw_dy = torch.randn(R, S, 4, generator=g) * torch.tensor([1, 1, 1, 3.0])
  raw_st = torch.randn(R, S, 4, generator=g) * torch.tensor([1, 1, 1, 3.0])
  raw_st[3, 10:20, 3] = -1e9   # masked-out samples (mlp_network.py:510-512)
  raw_dy[R - 1, :, 3] = 30.0   # softplus linear branch
  z = torch.sort(torch.rand(R, S, generator=g) * 10 + 0.5, dim=1)[0]
  pm_dy = (torch.rand(R, S, generator=g) > 0.7)
  pm_st = (torch.rand(R, S, generator=g) > 0.9)
  ref = O.raw2outputs(raw_dy, raw_st, z, pm_dy, pm_st)
  out = ops.composite(raw_dy.to(device), z.to(device), pm_dy.float().to(device), raw_st.to(device), pm_st.float().to(device))
  for k in ('rgb', 'rgb_static', 'rgb_dy', 'depth', 'alpha_dy', 'weights_dy', 'weights_st', 'alpha', 'weights'):
    assert_close(out[k], ref[k], 2e-6, 1e-5, f'raw2outputs {k}')
  assert_bitexact(out['mask'] > 0, ref['mask'], 'raw2outputs mask')
  ref = O.raw2outputs_vanilla(raw_dy, z, pm_dy)
  out = ops.composite(raw_dy.to(device), z.to(device), pm_dy.float().to(device))
  for k in ('rgb', 'depth', 'alpha', 'weights'):
    assert_close(out[k], ref[k], 2e-6, 1e-5, f'raw2outputs_vanilla {k}')
  assert_bitexact(out['mask'] > 0, ref['mask'], 'vanilla mask')


def check_fine_samples(device, golden, S=64):
  """Index-exactness protocol: weights come from the golden composite of the real reference; the kernel's
  sequential cdf must reproduce the reference's above_inds exactly and the sorted depths to 1 ulp-level tolerance."""
  z = torch.from_numpy(golden['composite/z_vals'])
  w = torch.from_numpy(golden['composite/weights'])
  R = z.shape[0]
  total_mismatch = 0
  for inv in (True, False):
    for mode in ('det', 'rand'):
      u = None if mode == 'det' else torch.from_numpy(golden[f'pdf/inv{int(inv)}/u'])
      z_all_r, inds_r = O.fine_z_vals(z, w, S, inv, mode == 'det', u, return_inds=True)
      z_all, z_s, inds = ops.fine_samples(z.to(device), w.to(device), S, inv, None if u is None else u.to(device), want_inds=True)
      mism = int((cpu(inds).long() != inds_r).sum())
      total_mismatch += mism
      assert mism == 0, f'fine samples inv={inv} {mode}: {mism} index mismatches'
      smp_ref = torch.from_numpy(golden[f'pdf/inv{int(inv)}/{mode}'])     # in sample_pdf's domain (1/z when inv)
      tie, tol = O.cdf_sample_conditioning(z, w, S, inv, mode == 'det', u)
      got = 1.0 / cpu(z_s) if inv else cpu(z_s)
      over = ((got - smp_ref).abs() > tol + 2e-6 * smp_ref.abs() + 1e-7) & ~tie
      assert int(over.sum()) == 0, (f'z_samples inv={inv} {mode}: {int(over.sum())} samples beyond the conditioning bound, '
                                    f'worst {float(((got - smp_ref).abs() - tol)[~tie].max()):.3e}')
      # the merge is pure data movement: bit-exact against a sort of the kernel's own new depths
      assert_bitexact(z_all, torch.sort(torch.cat([z, cpu(z_s)], dim=1), dim=1)[0], f'sorted union inv={inv} {mode}')
      clean = ~(tie.any(dim=1))
      assert_close(cpu(z_all)[clean], z_all_r[clean], 0.0, 0.02, f'z_all vs oracle (coarse sanity) inv={inv} {mode}')
      zs = cpu(z_all)
      assert bool((zs[:, 1:] >= zs[:, :-1]).all()), 'fine depths not sorted'
  return total_mismatch


def _weights(which):
  return cases.model_weights_trained() if which == 'trained' else cases.model_weights(0)


def static_inputs(name, S, R=None, weights='init'):
  """Oracle-side stage tensors of the static branch for a seeded scene."""
  scene, o, d, uv, _ = cases.scene_case(name)
  if R is not None:
    o, d = o[:R], d[:R]
  sd = O.tdict(_weights(weights)['net_coarse_st'])
  out, st = O.static_branch_pass(sd, scene, o, d, S, True, True, True, False, return_stages=True)
  return scene, o, d, sd, out, st


def check_static_net(device, name='small', S=64, R=None, aa=True, mask_rgb=False, atol=1e-4, weights='init', dark=0.0):
  """DynibarStatic on the oracle's own stage inputs (isolates the network kernels from the gather).
  dark > 0 (with mask_rgb): that fraction of the (point, view) rows -- and ALL views of a tenth as many points -- get a black source colour, so that
  mask_rgb (mlp_network.py:457-460) removes valid rows and whole points: the reference's masked_fill then sees the PRODUCT mask (a point whose every
  remaining row is dark blends uniformly over all V gathered colours, projection-masked ones included)."""
  scene, o, d, sd, _, st = static_inputs(name, S, R, weights)
  if dark > 0.0:
    g = torch.Generator().manual_seed(77)
    rf = st['rgb_feat'].clone()
    row_dark = torch.rand(rf.shape[:3], generator=g) < dark
    row_dark |= (torch.rand(rf.shape[:2], generator=g) < 0.1 * dark + 0.02)[..., None]
    rf[..., :3] = torch.where(row_dark[..., None], torch.zeros(()), rf[..., :3])
    st = dict(st, rgb_feat=rf)
  net_args = (sd, st['pts'], st['ref_rays_coords'], st['src_rays_coords'], st['rgb_feat'], F.normalize(d, dim=-1), st['ray_diff'], st['mask'])
  raw_ref = O.static_net(*net_args, aa, mask_rgb)
  # Conditioning of the reference itself: its anti-alias pooling weights are (e - min_v e) with e = exp(|s|(dot-1)) ~ 1, so one ulp
  # of exp() is amplified by 1/(spread of e over the views) (far samples: ~1e3).  `sens` = how far the oracle's own output moves
  # under +-1 ulp jitter of e; the tolerance is atol + 4 * sens per element (zero jitter sensitivity -> plain atol).
  sens = torch.zeros_like(raw_ref)
  if aa:
    for k in range(3):
      jit = (torch.randint(0, 3, st['mask'].shape, generator=torch.Generator().manual_seed(50 + k)).float() - 1.0) * 6e-8
      sens = torch.maximum(sens, (O.static_net(*net_args, aa, mask_rgb, exp_jitter=jit) - raw_ref).abs())
  sdev = to_dev(scene, device)
  views = ops.SourceViews(sdev['camera'], sdev['static_src_rgbs'], sdev['static_src_cameras'], sdev['static_featmaps'])
  net = ops.StaticNet(_weights(weights)['net_coarse_st'], device, aa, mask_rgb)
  raw = net(views, o.to(device), d.to(device), st['pts'].to(device), st['rgb_feat'].to(device), st['ray_diff'].to(device),
            st['mask'].to(device))
  sig, sig_ref = cpu(raw)[..., 3], raw_ref[..., 3]
  dead = sig_ref < -1e8
  assert bool((sig[dead] == sig_ref[dead]).all()), 'sigma of points without a valid view must be -1e9'
  err = (cpu(raw) - raw_ref).abs()
  lim = atol + 1e-4 * raw_ref.abs() + 4.0 * sens
  lim[..., 3][dead] = 0.0
  err[..., 3][dead] = 0.0
  live = ~dead
  record_margin(f'{name} static net sigma ({weights} weights, range {float(sig_ref[live].min()):.0f}..{float(sig_ref[live].max()):.0f})', err[..., 3][live], lim[..., 3][live])
  record_margin(f'{name} static net rgb ({weights} weights)', err[..., :3], lim[..., :3])
  over = err > lim
  assert int(over.sum()) == 0, (f'{name} static net: {int(over.sum())}/{err.numel()} outputs beyond atol {atol:.0e} + 4 x jitter sensitivity; '
                                f'worst excess {float((err - lim).max()):.3e}, max err rgb {float(err[..., :3].max()):.3e} sigma {float(err[..., 3].max()):.3e}')
  return float(err[..., :3].max()), float(err[..., 3].max()), float(sens.max())


def check_cross_axis(device, golden, name):
  """The shapes on which the reference's torch.cross WITHOUT dim (render_ray.py:375, :392) crosses over the views / the rays / the samples instead of
  xyz (csrc/dyn_device.h): the helper exports and DynibarStatic against the REAL reference's outputs on those shapes (tests/golden/cross_axis.npz),
  then the usual oracle checks of the network and of the whole static pass."""
  from dynibar_amd import render_ray as RR
  g = {k[len(name) + 1:]: v for k, v in golden.items() if k.startswith(name + '/')}
  S = cases.CROSS_AXIS_SAMPLES[name]
  scene, o, d, sd, _, st = static_inputs(name, S)
  dv = lambda x: x.to(device)
  assert_close(RR.compute_ref_plucker_coordinate(dv(o), dv(d)), torch.from_numpy(g['plucker/ref']), 1e-6, 1e-6, f'{name} ref Pluecker')
  assert_close(RR.compute_src_plucker_coordinate(dv(st['pts']), dv(scene['static_src_cameras'])), torch.from_numpy(g['plucker/src']), 2e-6, 1e-6,
               f'{name} src Pluecker')
  sdev = to_dev(scene, device)
  views = ops.SourceViews(sdev['camera'], sdev['static_src_rgbs'], sdev['static_src_cameras'], sdev['static_featmaps'])
  worst = 0.0
  for aa, mr in ((1, 0), (0, 1)):
    net = ops.StaticNet(_weights('init')['net_coarse_st'], device, aa, mr)
    raw = cpu(net(views, dv(o), dv(d), dv(st['pts']), dv(st['rgb_feat']), dv(st['ray_diff']), dv(st['mask'])))
    ref = torch.from_numpy(g[f'static_net/aa{aa}_mr{mr}/raw'])
    dead = ref[..., 3] < -1e8
    assert bool((raw[..., 3][dead] == ref[..., 3][dead]).all())
    err = (raw - ref).abs()
    err[..., 3][dead] = 0.0
    lim = 1e-4 + 1e-4 * ref.abs()
    if aa:  # the reference's own conditioning under +-1 ulp of exp() in the pooling weights (e - min_v e), as in check_static_net
      net_args = (sd, st['pts'], st['ref_rays_coords'], st['src_rays_coords'], st['rgb_feat'], F.normalize(d, dim=-1), st['ray_diff'], st['mask'])
      base = O.static_net(*net_args, True, bool(mr))
      for k in range(3):
        jit = (torch.randint(0, 3, st['mask'].shape, generator=torch.Generator().manual_seed(50 + k)).float() - 1.0) * 6e-8
        lim = torch.maximum(lim, 1e-4 + 1e-4 * ref.abs() + 4.0 * (O.static_net(*net_args, True, bool(mr), exp_jitter=jit) - base).abs())
    record_margin(f'{name} static net vs the reference itself (aa={aa}, mask_rgb={mr})', err, lim)
    assert int((err > lim).sum()) == 0, (f'{name} aa={aa} mask_rgb={mr}: max err rgb {float(err[..., :3].max()):.3e} sigma {float(err[..., 3].max()):.3e} '
                                         'against the reference\'s DynibarStatic')
    worst = max(worst, float(err[..., :3].max()))
  check_static_net(device, name, S=S)
  check_static_pass(device, name, S=S)
  return worst


def run_static_pass(device, scene_dev, net, o, d, S, inv_uniform=True, same_matrix=False):
  """BASELINE config 2 on the HIP path: sample -> project/gather -> DynibarStatic -> composite."""
  P = reference_proj_matrices(scene_dev['static_src_cameras'][0]) if same_matrix else None
  views = ops.SourceViews(scene_dev['camera'], scene_dev['static_src_rgbs'], scene_dev['static_src_cameras'], scene_dev['static_featmaps'], proj_matrices=P)
  R = o.shape[0]
  pts, z, s = ops.sample_along_ray(o, d, scene_dev['depth_range'], S, inv_uniform)
  rgb_feat, ray_diff, mask, pm = ops.project_gather(views, R, S, ray_o=o, ray_d=d, z_vals=z, pix_mask_thresh=1.0)
  raw = net(views, o, d, pts, rgb_feat, ray_diff, mask)
  return ops.composite(raw, z, pm), raw


def check_static_pass(device, name='small', S=64, R=None, atol=1e-4, weights='init', same_matrix=False):
  """same_matrix: the kernels start from the reference's own fp32 K.inv(c2w) (ops.SourceViews(proj_matrices=...)): then NO ray is dropped and NO
  conditioning allowance is added -- every ray, 1e-4."""
  scene, o, d, sd, out_ref, st = static_inputs(name, S, R, weights)
  net = ops.StaticNet(_weights(weights)['net_coarse_st'], device, True, False)
  out, raw = run_static_pass(device, to_dev(scene, device), net, o.to(device), d.to(device), S, same_matrix=same_matrix)
  # the anti-alias pooling weights (e - min_v e) are a cancellation (see check_static_net): 4 x how far +-1 ulp on exp() moves the ORACLE's own
  # outputs, per element -- negligible at initialisation scale, up to several 1e-4 with trained-scale density heads
  jit = {k: torch.zeros_like(out_ref[k]) for k in ('rgb', 'depth', 'weights')}
  Vs_ = scene['static_src_rgbs'].shape[1]
  for js in range(3):
    ej = (torch.randint(0, 3, (o.shape[0], S, Vs_, 1), generator=torch.Generator().manual_seed(50 + js)).float() - 1.0) * 6e-8
    oj = _oracle_static_graph(sd, scene, o, d, S, True, False, ej)[0]
    for k in jit:
      jit[k] = torch.maximum(jit[k], 4.0 * (oj[k] - out_ref[k]).abs())
  if same_matrix:
    tag = f'{name} static pass, reference matrices, all rays'
    assert_close(cpu(out['rgb']), out_ref['rgb'], atol, 0.0, f'{tag}: rgb', extra=jit['rgb'])
    assert_close(cpu(out['depth']), out_ref['depth'], 0.0, 2e-4, f'{tag}: depth', extra=jit['depth'])
    assert_close(cpu(out['weights']), out_ref['weights'], atol, 0.0, f'{tag}: weights', extra=jit['weights'])
    assert_bitexact(cpu(out['mask']) > 0, out_ref['mask'], f'{tag}: ray mask')
    return float((cpu(out['rgb']) - out_ref['rgb']).abs().max())
  # a sample whose projection sits on the frustum boundary may flip its mask (fp32 tie, see _mask_check); rays touching one are skipped
  Vs = scene['static_src_rgbs'].shape[1]
  margin = boundary_margin(st['pts'][None].repeat(Vs, 1, 1, 1), scene['static_src_cameras'][0]).any(dim=2).any(dim=1)
  keep = ~margin
  assert int(keep.sum()) > 0
  # white-noise maps amplify the last-bit differences of the projection matrices: measured on the oracle, per element
  sens = projection_sensitivity(lambda: O.static_branch_pass(sd, scene, o, d, S, True, True, True, False))
  ex = lambda k: SENS_FACTOR * sens[k][keep] + jit[k][keep]
  assert_close(cpu(out['rgb'])[keep], out_ref['rgb'][keep], atol, 0.0, f'{name} static pass rgb', extra=ex('rgb'))
  assert_close(cpu(out['depth'])[keep], out_ref['depth'][keep], 0.0, 2e-4, f'{name} static pass depth', extra=ex('depth'))
  assert_close(cpu(out['weights'])[keep], out_ref['weights'][keep], atol, 0.0, f'{name} static pass weights', extra=ex('weights'))
  assert_bitexact((cpu(out['mask']) > 0)[keep], out_ref['mask'][keep], f'{name} static pass ray mask')
  return float((cpu(out['rgb'])[keep] - out_ref['rgb'][keep]).abs().max())


def check_accuracy_against_double(device, name='small', S=64, R=None, weights='init', aa=True, mask_rgb=False):
  """How close is the HIP path to the EXACT values of the reference's formulas, next to how close the reference's own fp32 arithmetic is?  The same
  static pass (sample -> project / gather -> DynibarStatic -> composite) three ways: the oracle in float64 (exact for these inputs to 1e-15), the
  oracle in float32 (what the reference computes) and the kernels.  This check carries no tolerance chosen for the kernels: per output the kernels'
  error against the double values (largest, 99th and 90th percentile over the rays that touch no frustum boundary) is held to TWICE the fp32 reference's
  own error against them, plus 2e-6 of the output's scale for outputs the reference happens to get exact.  Returns the table."""
  scene, o, d, sd32, v32, _, _, keep = train_static_reference(name, S, R, aa, mask_rgb, weights)
  _, _, _, _, v64, _, _, keep64 = train_static_reference(name, S, R, aa, mask_rgb, weights, dtype=torch.float64)
  keep = keep & keep64
  assert int(keep.sum()) > 0
  net = ops.StaticNet(_weights(weights)['net_coarse_st'], device, aa, mask_rgb)
  out, raw = run_static_pass(device, to_dev(scene, device), net, o.float().to(device), d.float().to(device), S)
  ours = dict(raw=cpu(raw), rgb=cpu(out['rgb']), depth=cpu(out['depth']), weights=cpu(out['weights']))
  return _accuracy_table(f'{name}, {weights} weights, S={S}', ('rgb', 'weights', 'depth', 'raw'), ours, v32, v64, keep, sigma_in=('raw',))


def _accuracy_table(tag, keys, ours, v32, v64, keep, sigma_in=()):
  """per output: the kernels' and the fp32 oracle's absolute error against the float64 oracle (largest, 99th, 90th percentile over the kept rays); the
  kernels are held to twice the fp32 oracle's figures plus 2e-6 of the output's scale"""
  table = {}
  for k in keys:
    t64 = v64[k][keep]
    live = torch.ones_like(t64, dtype=torch.bool)
    if k in sigma_in:
      live = live & (t64[..., 3:4] > -1e8)  # points without a valid view: sigma is the constant -1e9 on all three sides (checked elsewhere)
    e_ref = ((v32[k][keep].double() - t64).abs())[live]
    e_our = ((ours[k][keep].double() - t64).abs())[live]
    scale = float(t64[live].abs().max())
    q = lambda e, p=0.99: float(torch.quantile(e.flatten()[:: max(1, e.numel() // 200000)], p))
    table[k] = dict(scale=scale, ref_max=float(e_ref.max()), ours_max=float(e_our.max()), ref_p99=q(e_ref), ours_p99=q(e_our), ref_p90=q(e_ref, 0.9), ours_p90=q(e_our, 0.9))
    floor = 2e-6 * max(scale, 1.0)
    record_margin(f'{tag} {k}: error against float64, kernels vs twice the fp32 reference\'s own (max)', torch.tensor([table[k]['ours_max']]),
                  torch.tensor([2.0 * table[k]['ref_max'] + floor]))
    # (a sample of a few dozen points -- the emulator's cases -- is held to its 90th percentile only: the largest of 32 heavy-tailed errors says nothing)
    for stat in (('max', 'p99', 'p90') if e_ref.numel() >= 2000 else ('p90',)):
      assert table[k]['ours_' + stat] <= 2.0 * table[k]['ref_' + stat] + floor, (
          f'{tag} {k} ({stat}): kernels {table[k]["ours_" + stat]:.3e} from the exact value, the fp32 reference {table[k]["ref_" + stat]:.3e}')
  print(f'  accuracy against float64 [{tag}]: ' + '; '.join(
      f'{k}: ours max {v["ours_max"]:.2e} p99 {v["ours_p99"]:.2e} | reference fp32 max {v["ref_max"]:.2e} p99 {v["ref_p99"]:.2e}' for k, v in table.items()))
  return table


def check_dual_accuracy_against_double(device, name='small', S=64, R=None, weights='init', shift=5.0):
  """The same for both branches together: gather at the (given) motion-displaced points -> DynibarDynamic, gather -> DynibarStatic, raw2outputs (the
  two-branch compositing) -- inference kernels against the float64 oracle, held to twice the fp32 oracle's own error."""
  di, v32, _, _, keep = train_dual_reference(name, S, R, weights, shift)
  _, v64, _, _, keep64 = train_dual_reference(name, S, R, weights, shift, dtype=torch.float64)
  keep = keep & keep64
  assert int(keep.sum()) > 0
  sc = to_dev(di['scene'], device)
  views_dy = ops.SourceViews(sc['camera'], sc['src_rgbs'], sc['src_cameras'], sc['featmaps'])
  views_st = ops.SourceViews(sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], sc['static_featmaps'])
  od, dd, pts, pts_seq, z = (di[k].to(device) for k in ('o', 'd', 'pts', 'pts_seq', 'z'))
  Rn = od.shape[0]
  rf, _, mk, pm_dy = ops.project_gather(views_dy, Rn, S, pts_st=pts, xyz=pts_seq, pix_mask_thresh=1.0)
  rfs, rds, mks, pm_st = ops.project_gather(views_st, Rn, S, ray_o=od, ray_d=dd, z_vals=z, pix_mask_thresh=1.0)
  raw_dy = ops.DynamicNet(_weights(weights)['net_coarse_dy'], device, shift=shift)(dd, pts, rf, mk, di['temb'].to(device))
  raw_st = ops.StaticNet(_weights(weights)['net_coarse_st'], device, True, False)(views_st, od, dd, pts, rfs, rds, mks)
  out = ops.composite(raw_dy, z, pm_dy, raw_static=raw_st, pix_mask_st=pm_st)
  ours = dict(raw_dy=cpu(raw_dy), rgb=cpu(out['rgb']), rgb_dy=cpu(out['rgb_dy']), weights=cpu(out['weights']), weights_dy=cpu(out['weights_dy']))
  return _accuracy_table(f'dual {name}, {weights} weights, S={S}', ('rgb', 'rgb_dy', 'weights', 'weights_dy', 'raw_dy'), ours, v32, v64, keep, sigma_in=('raw_dy',))


def check_static_pass_bench_shape_every_ray(device, R=4096, S=64, V=8, block=1024):
  """BASELINE configs[1] -- THE shape bench.py's headline is quoted on, its own scene (synthetic.make_scene(seed=0), 288 x 512 images, 72 x 128 x 32
  white-noise maps, 8 static views) and its own rank-0 rays (sample_pixels(100), 4096 of them) -- through ONE 4096-ray pass of the HIP path
  (sample -> gather -> DynibarStatic -> composite, the launch geometry of the bench step) against the oracle on EVERY ray.  The kernels start from
  the reference's own fp32 K.inv(c2w): no ray is dropped, no projection allowance.  The oracle walks the rays in blocks (rays are independent;
  a 4096-ray pass of the torch-CPU restatement needs tens of GB)."""
  from dynibar_amd import synthetic as syn
  sc = syn.make_scene(seed=0, H=288, W=512, V=V, F=32, n_static=V)
  scene = {k: cases.t(v) for k, v in sc.items()}
  o_np, d_np, _ = syn.pixel_rays(sc['camera'], syn.sample_pixels(100, 288, 512, R))
  o, d = cases.t(o_np), cases.t(d_np)
  w = syn.make_weights('static', 0, 32)
  sd = O.tdict(w)
  net = ops.StaticNet(w, device, True, False)
  out, _ = run_static_pass(device, to_dev(scene, device), net, o.to(device), d.to(device), S, same_matrix=True)
  out = {k: cpu(out[k]) for k in ('rgb', 'depth', 'weights', 'mask')}
  worst = 0.0
  for b0 in range(0, R, block):
    ob, db = o[b0:b0 + block], d[b0:b0 + block]
    ref = O.static_branch_pass(sd, scene, ob, db, S, True, True, True, False)
    jit = {k: torch.zeros_like(ref[k]) for k in ('rgb', 'depth', 'weights')}
    for js in range(2):  # +-1 ulp on exp() of the anti-alias pooling weights (see check_static_net), per element
      ej = (torch.randint(0, 3, (ob.shape[0], S, V, 1), generator=torch.Generator().manual_seed(50 + js)).float() - 1.0) * 6e-8
      oj = _oracle_static_graph(sd, scene, ob, db, S, True, False, ej)[0]
      for k in jit:
        jit[k] = torch.maximum(jit[k], 4.0 * (oj[k] - ref[k]).abs())
    tag = f'bench shape (configs[1]) static pass, rays {b0}..{b0 + ob.shape[0]}'
    sl = slice(b0, b0 + ob.shape[0])
    assert_close(out['rgb'][sl], ref['rgb'], 1e-4, 0.0, f'{tag}: rgb', extra=jit['rgb'])
    assert_close(out['depth'][sl], ref['depth'], 0.0, 2e-4, f'{tag}: depth', extra=jit['depth'])
    assert_close(out['weights'][sl], ref['weights'], 1e-4, 0.0, f'{tag}: weights', extra=jit['weights'])
    assert_bitexact(out['mask'][sl] > 0, ref['mask'], f'{tag}: ray mask')
    worst = max(worst, float((out['rgb'][sl] - ref['rgb']).abs().max()))
  return worst


def check_mlp_selftest(device, rows=1000):
  import ctypes
  from dynibar_amd import _lib
  g = torch.Generator().manual_seed(0)
  W = (torch.rand(64, 64, generator=g) - 0.5) * 0.4
  b = torch.rand(64, generator=g) - 0.5
  x = torch.randn(rows, 64, generator=g)
  xd = x.to(device)
  y = torch.full((rows, 64), float('nan'), device=device)
  buf = torch.zeros(2 * 3 * 4096, device=device)
  _lib.call('dyn_mlp_selftest', ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()), _lib.ptr(xd), _lib.ptr(y), rows, _lib.ptr(buf),
            _lib.stream_of(xd))
  ref = F.elu(F.linear(F.elu(F.linear(x, W, b)), W, b))
  terms = _lib.lib().dyn_mlp_split_terms()
  kind = _lib.lib().dyn_mlp_split_kind()
  # two chained 64-wide layers on N(0,1) inputs (|y| up to ~2): fp32-class engines -- native fp32 MFMA, 6-term bf16 split and the
  # shipped 3-term half-float split (22-bit operands) -- hold 3e-6; the 3-term bf16 split (16-bit operands, products good to 2^-16)
  # holds 2.5e-5 (measured 2.0e-5)
  lim = 2.5e-5 if (terms == 3 and kind == 1) else 3e-6
  assert_close(y, ref, lim, 0.0, f'mlp engine self-test (split terms {terms}, kind {kind})')


def check_mlp_selftest_ranges(device, rows=512):
  """The inference engine at the edges of the half-float split's range (dyn_mlp.h: DYN_SPLIT_F16), through the same two chained 64-wide layers as the
  self-test, against an fp64 reference:
    * activations of 1e-6 .. 6e-5: the first part of such an operand is a SUBNORMAL half (an absolute grid of 2^-24 = 6e-8), the second part is below
      that grid: the operand is good to +-3e-8 absolute, not to 2^-22 relative -- fp32's own epsilon at O(1), invisible next to an O(1) bias;
    * activations of 7e4 .. 1.2e5: beyond the largest half (65504) the truncating convert saturates and the second part absorbs the rest with 11 bits:
      the product degrades gracefully to ~2^-12 relative instead of overflowing -- up to 2 x 65504; an activation beyond 131008 overflows the second
      part too and the result is not finite (measured in round 5: the weights of the `huge` case are scaled so that the hidden layer stays inside).
  THE RANGE OF THE ENGINE is therefore |activation| < 65504 at full precision, < 131008 degraded, undefined beyond -- for every value that enters a
  Linear layer, the weighted variances over the views included (features up to ~250 in magnitude).  Returns the measured errors (margin table)."""
  import ctypes
  from dynibar_amd import _lib
  terms, kind = _lib.lib().dyn_mlp_split_terms(), _lib.lib().dyn_mlp_split_kind()
  g = torch.Generator().manual_seed(3)
  W0 = (torch.rand(64, 64, generator=g) - 0.5) * 0.4
  b = torch.rand(64, generator=g) - 0.5
  out = {}
  for tag, lo, hi, wscale in (('tiny', 1e-6, 6e-5, 1.0), ('huge', 7e4, 1.2e5, 0.15)):
    W = (W0 * wscale).contiguous()
    mag = lo + (hi - lo) * torch.rand(rows, 64, generator=g)
    x = mag * torch.where(torch.rand(rows, 64, generator=g) < 0.5, -1.0, 1.0)
    xd = x.to(device)
    y = torch.full((rows, 64), float('nan'), device=device)
    buf = torch.zeros(2 * 3 * 4096, device=device)
    _lib.call('dyn_mlp_selftest', ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(b.data_ptr()), _lib.ptr(xd), _lib.ptr(y), rows, _lib.ptr(buf), _lib.stream_of(xd))
    Wd, bd = W.double(), b.double()
    h1 = F.elu(F.linear(x.double(), Wd, bd))
    ref = F.elu(F.linear(h1, Wd, bd))
    scale = F.linear(h1.abs(), Wd.abs(), bd.abs())  # sum |w| |h| + |b|: what the products of the last layer are made of
    err = (cpu(y).double() - ref).abs()
    assert float(h1.abs().max()) < 131008.0, f'{tag}: the test itself left the stated range (hidden activation {float(h1.abs().max()):.3g})'
    assert bool(torch.isfinite(cpu(y)).all()), f'{tag} activations: non-finite output'
    if tag == 'tiny' or not (terms == 3 and kind == 2):
      lim = torch.full_like(err, 3e-6) if tag == 'tiny' else 3e-6 * scale
    else:
      lim = 6e-4 * scale  # two half parts past the half range: 2^-11 relative on the part of the operand beyond 65504
    record_margin(f'mlp engine, activations {lo:g}..{hi:g} (hidden up to {float(h1.abs().max()):.3g}; split terms {terms}, kind {kind}); worst relative to '
                  f'sum|w||h| {float((err / scale).max()):.2e}', err, lim)
    assert bool((err <= lim).all()), f'{tag} activations: max err {float(err.max()):.3e}, relative to sum|w||h| {float((err / scale).max()):.3e}'
    out[tag] = float((err / scale).max())
  return out


def check_static_net_feature_range(device, scale, name='small', S=16, rtol=None):
  """One k_static_views / k_net_points / k_static_blend pass whose gathered FEATURE channels (the 32 learned channels; colours stay in [0,1]) are scaled
  towards the edges of the engine's range: x 3e-5 puts them at 1e-6 .. 6e-5 (subnormal first parts); x 60 puts them at up to ~250, where their weighted
  VARIANCE over the views -- an input of base_fc.0 like any other -- reaches the largest half (6e4): the upper edge of the range in terms of features.
  Against the oracle in fp64 on the same scaled inputs, the limit relative to the raw outputs' own magnitude plus what fp32 itself costs there."""
  scene, o, d, sd, _, st = static_inputs(name, S)
  rgb_feat = st['rgb_feat'].clone()
  rgb_feat[..., 3:] *= scale
  net_args = (sd, st['pts'], st['ref_rays_coords'], st['src_rays_coords'], rgb_feat, F.normalize(d, dim=-1), st['ray_diff'], st['mask'])
  raw_ref = O.static_net(*net_args, True, False)
  raw_ref64 = O.static_net({k: v.double() for k, v in sd.items()}, *[a.double() for a in net_args[1:]], True, False)
  sdev = to_dev(scene, device)
  views = ops.SourceViews(sdev['camera'], sdev['static_src_rgbs'], sdev['static_src_cameras'], sdev['static_featmaps'])
  net = ops.StaticNet(_weights('init')['net_coarse_st'], device, True, False)
  raw = cpu(net(views, o.to(device), d.to(device), st['pts'].to(device), rgb_feat.to(device), st['ray_diff'].to(device), st['mask'].to(device))).double()
  dead = raw_ref64[..., 3] < -1e8
  live = ~dead
  assert bool(torch.isfinite(raw).all()), f'feature scale {scale:g}: non-finite output'
  # fp32 oracle vs fp64 oracle = what fp32 arithmetic itself costs on these inputs; the kernels get the stated tolerance on top of it
  own = (raw_ref.double() - raw_ref64).abs()
  rtol = (1e-4 if scale < 1 else 2e-3) if rtol is None else rtol  # (x 60: variances of ~6e4 carry 22 bits here, 24 in the reference: measured 9e-4 on sigma)
  sig_scale = raw_ref64[..., 3][live].abs().max().clamp(min=1.0)
  err_rgb = (raw[..., :3] - raw_ref64[..., :3]).abs()
  err_sig = (raw[..., 3] - raw_ref64[..., 3]).abs()[live]
  lim_rgb = 1e-4 + (rtol if scale > 1 else 0.0) + 4 * own[..., :3]
  lim_sig = rtol * sig_scale + 1e-4 + 4 * own[..., 3][live]
  record_margin(f'static net, feature channels x {scale:g}: rgb', err_rgb, lim_rgb)
  record_margin(f'static net, feature channels x {scale:g}: sigma (scale {float(sig_scale):.3g})', err_sig, lim_sig)
  assert bool((err_rgb <= lim_rgb).all()), f'feature scale {scale:g}: rgb err {float(err_rgb.max()):.3e}'
  assert bool((err_sig <= lim_sig).all()), f'feature scale {scale:g}: sigma err {float(err_sig.max()):.3e} of {float(sig_scale):.3g}'
  return float(err_rgb.max()), float((err_sig / sig_scale).max())


def dynamic_inputs(name, S, R=None, weights='init'):
  """Oracle-side stage tensors of the dynamic branch (motion MLP -> trajectory points -> projection) for a seeded scene."""
  scene, o, d, uv, _ = cases.scene_case(name)
  if R is not None:
    o, d = o[:R], d[:R]
  W = {k: O.tdict(v) for k, v in _weights(weights).items()}
  basis = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  pts, z, s = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  Rn = pts.shape[0]
  t_emb = temb[None, None, :].repeat(Rn, S, 1)
  coeff = O.motion_mlp(W['motion_mlp'], torch.cat([pts, t_emb], -1).float())
  n_last = int(round(S * 0.1))
  coeff[:, -n_last:, :] *= 0.0
  traj = O.trajectory_points(coeff, basis, fidx)
  pts_seq = torch.stack([pts + (traj[k] - traj[0]) for k in toff], 0)
  rf, rd, mk = O.compute_with_motions(pts, pts_seq, scene['camera'], scene['src_rgbs'], scene['src_cameras'], scene['featmaps'])
  return dict(scene=scene, o=o, d=d, W=W, basis=basis, pts=pts, z=z, fidx=fidx, temb=temb, toff=toff, t_emb=t_emb, coeff=coeff,
              pts_seq=pts_seq, rgb_feat=rf, ray_diff=rd, mask=mk, n_last=n_last)


def check_motion(device, name='small', S=64, R=None, weights='init'):
  di = dynamic_inputs(name, S, R, weights)
  mm = ops.MotionMLP(_weights(weights)['motion_mlp'], device, cases.NUM_BASIS)
  coeff = mm(di['pts'].to(device), di['temb'].to(device), di['n_last'])
  # 8 ReLU layers of width 256 on Fourier features up to 17 * |x|: fp32 round-off grows with the argument of sin/cos
  assert_close(coeff, di['coeff'], 2e-5, 1e-4, f'{name} motion coefficients')
  rows = [di['fidx'] + k for k in di['toff']]
  seq = ops.trajectory_points(di['coeff'].to(device), di['basis'].to(device), di['pts'].to(device), rows, di['fidx'])
  assert_close(seq, di['pts_seq'], 1e-6, 1e-6, f'{name} trajectory points')
  return float((cpu(coeff) - di['coeff']).abs().max())


def check_fused_trajectory(device, name='small', S=64, R=None, weights='init', virtual_views=0):
  """compute_traj_pts fused into its consumers (render_ray.py:361-369, :691-725): the gather and the flows that form the displaced points themselves from
  the motion coefficients must equal -- the gather bit for bit, the flows to 5e-5 px -- the gather / flows on the materialised [V,R,S,3] array of k_trajectory_points (same sums, same order:
  csrc/dyn_geometry.hip traj_displace), with scaled-up coefficients so that the displacement moves the taps, and with virtual views (row < 0: no motion)."""
  di = dynamic_inputs(name, S, R, weights)
  sc = to_dev(di['scene'], device)
  views = ops.SourceViews(sc['camera'], sc['src_rgbs'], sc['src_cameras'], sc['featmaps'])
  V = views.V
  pts = di['pts'].to(device)
  coeff = (di['coeff'] * 40.0).to(device).contiguous()  # (initialisation-scale motion is ~1e-3 of the scene: scaled so that pixels and masks really move)
  basis = di['basis'].to(device)
  rows = [di['fidx'] + k for k in di['toff']]
  if virtual_views:
    rows = rows[:V - virtual_views] + [-1] * virtual_views
  assert len(rows) == V
  ref = di['fidx']
  Rn = pts.shape[0]
  rows_dev = torch.tensor(rows, dtype=torch.int32, device=device)
  seq = ops.trajectory_points(coeff, basis, pts, rows, ref)
  assert float((seq - pts[None]).abs().max()) > 1e-2, 'the case does not displace anything'
  a = ops.project_gather(views, Rn, S, pts_st=pts, xyz=seq, pix_mask_thresh=1.0)
  b = ops.project_gather(views, Rn, S, pts_st=pts, pix_mask_thresh=1.0, traj=(coeff, basis, rows_dev, ref))
  for x, y, what in zip(a, b, ('rgb_feat', 'ray_diff', 'mask', 'pix_mask')):
    assert_bitexact(cpu(y), cpu(x), f'{name} fused trajectory gather: {what}')
  static = ops.project_gather(views, Rn, S, pts_st=pts, pix_mask_thresh=1.0)
  assert float((static[0] - a[0]).abs().max()) > 1e-3, 'the displaced gather equals the undisplaced one: the case pins nothing'
  g = torch.Generator().manual_seed(3)
  w = torch.rand(Rn, S, generator=g)
  w = (w / w.sum(dim=1, keepdim=True)).to(device).contiguous()
  uv = torch.rand(Rn, 2, generator=g).mul(40.0).to(device).contiguous()
  fa = torch.empty((V, Rn, 2), dtype=torch.float32, device=device)
  fb = torch.empty_like(fa)
  ops.call('dyn_render_flows', ops.ptr(w), ops.ptr(seq), ops.ptr(views.proj), ops.ptr(uv), Rn, S, V, ops.ptr(fa), ops.stream_of(fa))
  ops.call('dyn_render_flows_traj', ops.ptr(w), ops.ptr(pts.contiguous()), ops.ptr(coeff), ops.ptr(basis.contiguous()), int(basis.shape[1]), rows_dev.data_ptr(), int(ref),
           ops.ptr(views.proj), ops.ptr(uv), Rn, S, V, ops.ptr(fb), ops.stream_of(fb))
  # (the flows form the expected point through its linearity in the coefficients -- another order of the fp32 sums: 1e-7 of the scene's scale, amplified by the
  # projection to ~1e-5 px; the reference's own flows are compared at 2e-4 px)
  assert_close(cpu(fb), cpu(fa), 5e-5, 1e-5, f'{name} fused trajectory flows')
  return float((seq - pts[None]).abs().max())


def check_expected_scene_flow(device, R=37, S=150, B=6, seed=4):
  """k_expected_scene_flow (render_ray.py:584-595 / :1086-1096) on random weights and coefficients, rays longer than one 64-sample staging block and a ray
  count that is no multiple of a workgroup's four, against torch in double."""
  g = torch.Generator().manual_seed(seed)
  w = torch.rand(R, S, generator=g)
  w = w / w.sum(dim=1, keepdim=True)
  coeff = torch.randn(R, S, 3 * B, generator=g)
  basis = O.init_dct_basis(B, cases.NUM_FRAMES)
  rp, rm, rr = 12, 10, 11
  out = torch.empty((R, 3), dtype=torch.float32, device=device)
  wd, cd, bd = w.to(device).contiguous(), coeff.to(device).contiguous(), basis.to(device).contiguous()
  ops.call('dyn_expected_scene_flow', ops.ptr(wd), ops.ptr(cd), ops.ptr(bd), R, S, B, rp, rm, rr, ops.ptr(out), ops.stream_of(out))
  c = coeff.double().reshape(R, S, 3, B)
  tr = lambda row: (c * basis[row].double()).sum(-1)
  ref = torch.max((w.double()[..., None] * (tr(rp) - tr(rr))).sum(1), (w.double()[..., None] * (tr(rm) - tr(rr))).sum(1))
  assert_close(cpu(out).double(), ref, 2e-6, 1e-5, 'expected scene flow')
  return float((cpu(out).double() - ref).abs().max())


def check_dynamic_net(device, name='small', S=64, R=None, shift=0.0, atol=1e-4, weights='init'):
  di = dynamic_inputs(name, S, R, weights)
  Vd = di['rgb_feat'].shape[2]
  tdiff = torch.zeros(di['pts'].shape[0], S, Vd, 1)
  raw_ref = O.dynamic_net(di['W']['net_coarse_dy'], di['pts'], di['rgb_feat'], F.normalize(di['d'], dim=-1), di['ray_diff'], tdiff, di['mask'],
                          di['t_emb'], shift=shift)
  net = ops.DynamicNet(_weights(weights)['net_coarse_dy'], device, shift=shift)
  raw = net(di['d'].to(device), di['pts'].to(device), di['rgb_feat'].to(device), di['mask'].to(device), di['temb'].to(device))
  sig, sig_ref = cpu(raw)[..., 3], raw_ref[..., 3]
  dead = sig_ref < -1e8
  assert bool((sig[dead] == sig_ref[dead]).all()), 'sigma of points without a valid view must be -1e9'
  assert_close(sig[~dead], sig_ref[~dead], atol, 1e-4, f'{name} dynamic sigma ({weights} weights, range {float(sig_ref[~dead].min()):.0f}..{float(sig_ref[~dead].max()):.0f})')
  assert_close(cpu(raw)[..., :3], raw_ref[..., :3], atol, 0.0, f'{name} dynamic rgb ({weights} weights)')
  return float((cpu(raw) - raw_ref)[..., :3].abs().max()), float((sig[~dead] - sig_ref[~dead]).abs().max())


# ----------------------------------------------------------------------------------------------------------------------
# the drop-in layer (dynibar_amd.sample_ray / projection / render_ray / render_image) against the golden outputs of the REAL reference
# ----------------------------------------------------------------------------------------------------------------------
def make_model(device, seed=0):
  """An object with the attribute names of the reference's DynibarFF (model.py:33-101) holding plain state dicts."""
  import types
  W = cases.model_weights(seed)
  m = types.SimpleNamespace(**W)
  m.trajectory_basis = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES).to(device)
  m.trajectory_basis_fine = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES).to(device)
  return m


def make_ray_batch(scene, o, d, uv, device):
  b = {k: scene[k].to(device) for k in ('camera', 'src_rgbs', 'src_cameras', 'static_src_rgbs', 'static_src_cameras', 'depth_range')}
  b.update(ray_o=o.to(device), ray_d=d.to(device), uv_grid=uv.to(device))
  return b


def _group_tol(key, name=None):
  # per-sample probabilities and rendered colours: 1e-4 (north_star); depths scale with the scene; flows are pixel differences.
  # No blanket loosening for ill-conditioned scenes: their allowance is measured per element (projection_sensitivity).
  if key in ('depth', 'z_vals', 's_vals'):
    return dict(atol=2e-4, rtol=2e-4)
  if key in ('render_flows',):  # pixel differences of projected expected points (values up to the image width): measured 1.5e-5 px at worst
    return dict(atol=2e-4, rtol=1e-5)
  if key in ('exp_sf',):        # measured 6e-8
    return dict(atol=1e-6, rtol=1e-4)
  return dict(atol=1e-4, rtol=1e-4)


SENS_FACTOR = 3.0


def projection_sensitivity(run):
  """Conditioning of the REFERENCE algorithm with respect to the last bits of K.inv(c2w): the kernels form that matrix in double
  (k_prepare_cameras), the reference with an fp32 LU inverse + fp32 bmm (projection.py:42-47); neither is "the" fp32 answer.  `run`
  evaluates the oracle; it is evaluated once as the reference does and once with the double-formed matrix, and the per-element
  difference of its outputs (nested dicts of tensors) is how far the reference's own result moves under that perturbation.  On smooth
  maps it is ~1e-7; on white-noise maps (gradient ~1 per pixel) a 1e-4 px shift of a tap is visible.  Checks add SENS_FACTOR x this."""
  base = run()
  O.PROJECTION_MODE = 'double'
  try:
    pert = run()
  finally:
    O.PROJECTION_MODE = 'reference'

  def diff(a, b):
    if isinstance(a, dict):
      return {k: diff(a[k], b[k]) for k in a if a[k] is not None}
    if isinstance(a, torch.Tensor) and a.dtype.is_floating_point:
      return (a.double() - b.double()).abs()
    return None
  return diff(base, pert)


def check_group_vs_golden(prefix, out, golden, name, skip_rays=None, sens=None):
  n = 0
  for k, v in out.items():
    if v is None:
      continue
    ref = torch.from_numpy(golden[prefix + k])
    got = cpu(v)
    extra = None
    if sens is not None and sens.get(k) is not None:
      extra = SENS_FACTOR * sens[k]
    if skip_rays is not None and skip_rays.any():
      ax = 1 if got.dim() == 3 and got.shape[1] == skip_rays.numel() else 0
      keep = ~skip_rays
      got, ref = (got[:, keep], ref[:, keep]) if ax == 1 else (got[keep], ref[keep])
    if ref.dtype == torch.bool:
      assert_bitexact(got, ref, prefix + k)
    else:
      tol = _group_tol(k, name)
      assert_close(got.float(), ref.float(), tol['atol'], tol['rtol'], f'{name}:{prefix}{k}', extra=extra)
    n += 1
  return n


def oracle_models():
  W = {k: O.tdict(v) for k, v in cases.model_weights(0).items()}
  W['trajectory_basis'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  W['trajectory_basis_fine'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  return W


def check_render_rays_mv(device, golden, name='small', S=64, same_matrix=False):
  """render_rays_mv (coarse 64 + fine 64, dynamic + static, det=True, inv_uniform=True) against the real reference's outputs.
  same_matrix: the projector is handed the reference's own fp32 K.inv(c2w) (Projector(matrix_mode=...)); no conditioning allowance is added then,
  not even on the white-noise scene."""
  import types
  from dynibar_amd import projection, render_ray
  scene, o, d, uv, _ = cases.scene_case(name)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  proj = projection.Projector(device, matrix_mode=reference_proj_matrices) if same_matrix else projection.Projector(device)
  batch = make_ray_batch(scene, o, d, uv, device)
  cfeat = (scene['featmaps'].to(device), None, scene['static_featmaps'].to(device))
  ffeat = (scene['featmaps_fine'].to(device), None, scene['static_featmaps_fine'].to(device))
  ret = render_ray.render_rays_mv((fidx, None), (temb.to(device), None), (toff, None), batch, model, proj, cfeat, ffeat, S, args,
                                  inv_uniform=True, N_importance=S, det=True, is_train=False)
  sens = {}
  if name == 'noise' and not same_matrix:  # white-noise maps: the reference's own conditioning w.r.t. the projection matrices' last bits, per element
    W = oracle_models()
    sens = projection_sensitivity(lambda: O.render_rays_mv(W, dict(scene), o, d, uv, fidx, temb, toff, S, S))
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_fine_ref', 'outputs_fine_ref_dy'):
    n += check_group_vs_golden(f'mv/{grp}/' , ret[grp], golden, name + (' [reference matrices]' if same_matrix else ''), sens=sens.get(grp))
  assert n == sum(1 for k in golden if k.startswith('mv/')), 'render_rays_mv output key set differs from the reference'
  assert ret['outputs_fine_anchor'] is None and ret['outputs_fine_anchor_dy'] is None
  # chain-level index exactness (protocol b): the inverse-CDF indices the HIP chain derives from ITS OWN coarse
  # weights against the indices the real reference derived from its own (recorded inside the reference's render_rays_mv)
  if 'chain/mv_above_inds' in golden:
    c = ret['outputs_coarse_ref']
    _, _, inds = ops.fine_samples(c['z_vals'], c['weights'], S, True, None, want_inds=True)
    ref_inds = torch.from_numpy(golden['chain/mv_above_inds']).long()
    mism = cpu(inds).long() != ref_inds
    n_mis = int(mism.sum())
    CHAIN_INDEX_REPORT.append(dict(case=name, samples=int(ref_inds.numel()), mismatches=n_mis))
    if n_mis:
      # A flip is legitimate only where u sits within the cdf's own uncertainty of every knot it jumped: the reference's index is then
      # decided by the last bits of ITS coarse weights, which carry the network round-off of either implementation.
      w_ref = torch.from_numpy(golden['mv/outputs_coarse_ref/weights'])
      w_err = float((cpu(c['weights']) - w_ref).abs().max())
      ww = torch.flip(w_ref[:, 1:-1], dims=[1])  # inv_uniform=True ordering (render_ray.py:793-797)
      cdf = O.pdf_to_cdf(ww.clone())
      bound = 4e-7 + 2.0 * w_err * ww.shape[1] / float((ww + 1e-5).sum(dim=1).min())
      u = torch.linspace(0.0, 1.0, S)
      got = cpu(inds).long()
      unexplained = 0
      for r, k in zip(*torch.nonzero(mism, as_tuple=True)):
        lo, hi = sorted((int(got[r, k]), int(ref_inds[r, k])))
        if float((cdf[r, lo:hi] - u[k]).abs().max()) > bound:
          unexplained += 1
      CHAIN_INDEX_REPORT[-1].update(knot_ties=n_mis - unexplained, tie_bound=bound)
      assert unexplained == 0, f'{name}: {unexplained} inverse-CDF index flips that are not knot ties (of {n_mis} flips, {ref_inds.numel()} samples)'
  return n


CHAIN_INDEX_REPORT = []


def check_render_rays_mono(device, golden, name='small', S=64):
  import types
  from dynibar_amd import projection, render_ray
  scene, o, d, uv, _ = cases.scene_case(name)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  proj = projection.Projector(device)
  batch = make_ray_batch(scene, o, d, uv, device)
  feat = (scene['featmaps'].to(device), None, scene['static_featmaps'].to(device))
  ret = render_ray.render_rays_mono((fidx, None), (temb.to(device), None), (toff, None), batch, model, feat, proj, S, args, inv_uniform=True,
                                    det=True, is_train=False, num_vv=0)
  sens = {}
  if name == 'noise':
    W = oracle_models()
    sens = projection_sensitivity(lambda: O.render_rays_mono_eval(W, dict(scene), o, d, uv, fidx, temb, toff, S, True, True, num_vv=0))
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_coarse_ref_dy', 'outputs_coarse_st'):
    n += check_group_vs_golden(f'mono/{grp}/', ret[grp], golden, name, sens=sens.get(grp))
  assert n == sum(1 for k in golden if k.startswith('mono/')), 'render_rays_mono output key set differs from the reference'
  return n


def make_module_model(device, args, shift=5.0, wrap=True, weights=None):
  """A DynibarMono-shaped model whose nets are real nn.Modules on `device`, DataParallel-wrapped like model.py:382-397, with the
  conditional parameter set of the reference's constructors for `args` (no `s` when anti_alias_pooling = 0)."""
  import types
  import refmodules
  W = cases.model_weights(0) if weights is None else weights
  dp = (lambda m: torch.nn.DataParallel(m.to(device))) if wrap else (lambda m: m.to(device))
  m = types.SimpleNamespace()
  m.net_coarse_st = dp(refmodules.load_numpy_state(refmodules.like_reference('static', args), W['net_coarse_st']))
  m.net_coarse_dy = dp(refmodules.load_numpy_state(refmodules.like_reference('dynamic', args, shift=shift), W['net_coarse_dy']))
  m.motion_mlp = dp(refmodules.load_numpy_state(refmodules.like_reference('motion', args, num_basis=cases.NUM_BASIS), W['motion_mlp']))
  m.trajectory_basis = torch.nn.parameter.Parameter(O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)).float().to(device).detach().requires_grad_(True)
  return m


def check_render_rays_mono_kid(device, golden, S=64):
  """render_rays_mono with the arguments every monocular config ships (configs/train_kid-running.txt:41-42,69: anti_alias_pooling=0,
  mask_rgb=1, num_vv=3) on DataParallel-wrapped nn.Modules WITHOUT an `s` parameter, against the real reference's outputs."""
  import types
  from dynibar_amd import projection, render_ray
  scene, o, d, uv, _ = cases.scene_case('kid')
  fidx, temb, toff = cases.time_args(7)
  args = types.SimpleNamespace(anti_alias_pooling=0, mask_rgb=1, input_dir=True, input_xyz=False, occ_weights_mode=0)
  model = make_module_model(device, args, shift=5.0)
  assert not any(k.endswith('s') and '.' not in k.replace('module.', '') for k in model.net_coarse_st.state_dict()), 'stand-in must not own `s`'
  batch = make_ray_batch(scene, o, d, uv, device)
  feat = (scene['featmaps'].to(device), None, scene['static_featmaps'].to(device))
  ret = render_ray.render_rays_mono((fidx, None), (temb.to(device), None), (toff, None), batch, model, feat, projection.Projector(device), S, args,
                                    inv_uniform=True, det=True, is_train=False, num_vv=3)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_coarse_ref_dy', 'outputs_coarse_st'):
    n += check_group_vs_golden(f'mono/{grp}/', ret[grp], golden, 'kid')
  assert n == len(golden), 'render_rays_mono (kid-running arguments) output key set differs from the reference'
  return n


def check_encoder(device, golden, name='small', atol=1e-4):
  """dynibar_amd.feature_network.ResNet (HIP convolutions, channels-last) against the real reference's ResNet outputs: 1e-4 (+1e-4
  relative); returned maps are NCHW views of channels-last storage that SourceViews taps in place."""
  from dynibar_amd import feature_network
  imgs, sd = cases.encoder_case(name)
  net = feature_network.ResNet.from_module(sd)
  x = imgs.to(device).permute(0, 3, 1, 2)  # what eval_nvidia.py:335-358 passes: an NCHW view of [N,H,W,3]
  xc, xf = net(x)
  for got, key in ((xc, 'coarse'), (xf, 'fine')):
    ref = torch.from_numpy(golden[f'{name}/{key}'])
    assert tuple(got.shape) == tuple(ref.shape), f'{key}: shape {tuple(got.shape)} vs reference {tuple(ref.shape)}'
    assert ops._channels_last_view(got) is not None, 'encoder outputs must be channels-last in memory'
    assert_close(got, ref, atol, 1e-4, f'encoder {name} {key} (|ref| up to {float(ref.abs().max()):.1f})')
  # the oracle's restatement on the same inputs (pinned to the same golden by tests/test_oracle_golden.py)
  oc, of = O.resnet_encoder(O.tdict(sd), imgs.permute(0, 3, 1, 2))
  assert_close(xc, oc, atol, 1e-4, f'encoder {name} coarse vs oracle')
  return float((cpu(xc) - torch.from_numpy(golden[f'{name}/coarse'])).abs().max())


def encoder_graph_with_masks(p, x, masks=None):
  """The executed part of ResNet.forward exactly as oracle/ibr_oracle.py:resnet_encoder states it (feature_network.py:302-311), with the
  ReLUs optionally replaced by GIVEN 0/1 masks -> (coarse, fine, pre-activations of the four kinds of ReLU in order).  With masks=None it
  is the oracle's function (asserted by the caller); with the product's own masks it is the smooth function whose gradient the product
  must reproduce -- a ReLU whose argument is within rounding of zero is not determined by the reference's fp32 arithmetic either."""
  pre_acts, it = [], iter(masks) if masks is not None else None

  def relu(v):
    pre_acts.append(v)
    return F.relu(v) if it is None else v * next(it)

  x = relu(O._inorm(p, 'bn1', O._conv_reflect(x, p['conv1.weight'], 2, 3)))
  for b in range(3):
    pre = 'layer1.%d.' % b
    identity = x
    out = relu(O._inorm(p, pre + 'bn1', O._conv_reflect(x, p[pre + 'conv1.weight'], 2 if b == 0 else 1, 1)))
    out = O._inorm(p, pre + 'bn2', O._conv_reflect(out, p[pre + 'conv2.weight'], 1, 1))
    if b == 0:
      identity = O._inorm(p, pre + 'downsample.1', O._conv_reflect(x, p[pre + 'downsample.0.weight'], 2, 0))
    x = relu(out + identity)
  x_out = F.conv2d(x, p['out_conv.weight'], p['out_conv.bias'])
  return x_out[:, :32], x_out[:, -32:], pre_acts


def check_encoder_training(device, name='small'):
  """Training form of the feature encoder (dynibar_amd.train_encoder) against autograd through the oracle's restatement of the reference's
  ResNet (pinned to the reference's own outputs by the encoder golden): the maps, and the gradient of a seeded linear functional of both
  maps w.r.t. every parameter the executed part of ResNet.forward has (feature_network.py:179-311).  ReLU arguments within rounding of zero
  (one or two of 10^5 on a GPU, none under the emulator) make the gradient jump: the reference gradient is taken with the product's own
  ReLU decisions, after checking that they differ from the oracle's only where the oracle's argument is |v| < 1e-5 of the map's scale.
  Also: a ResNet wrapper around a module with trainable parameters takes this path under grad mode and the forward-only kernels under
  no_grad, with the same values."""
  from dynibar_amd import feature_network, train_encoder
  imgs, sd = cases.encoder_case(name)
  g = torch.Generator().manual_seed(3)
  x64 = imgs.permute(0, 3, 1, 2).double()
  ref_p = {k: v.clone().double().requires_grad_(True) for k, v in O.tdict(sd).items() if k in train_encoder.PARAMS}
  with torch.no_grad():
    oc0, of0 = O.resnet_encoder(ref_p, x64)
    oc1, of1, pre_oracle = encoder_graph_with_masks(ref_p, x64)
  assert torch.equal(oc0, oc1) and torch.equal(of0, of1), 'the masked restatement must BE the oracle function when no masks are given'
  # the product: forward with its saved state (for the ReLU decisions), then the public entry for the gradients
  w = {k: torch.from_numpy(np.asarray(sd[k])).float().to(device) for k in train_encoder.PARAMS}
  _, _, state = train_encoder._forward(w, imgs.to(device).contiguous())
  stem_x = state['stem'][3]
  relu_outs = [stem_x] + [t for blk in state['blocks'] for t in (blk[4], blk[9])]  # x0, then per block h1 and the block output
  masks = [(cpu(t).permute(0, 3, 1, 2) > 0).double() for t in relu_outs]
  flips = 0
  for m, v in zip(masks, pre_oracle):
    diff = (m > 0) != (v > 0)
    flips += int(diff.sum())
    if bool(diff.any()):
      assert float(v[diff].abs().max()) < 1e-5 * float(v.abs().max()), 'a ReLU decision differs from the oracle away from zero'
  assert flips <= 8, f'{flips} ReLU decisions differ from the oracle'
  oc, of, _ = encoder_graph_with_masks(ref_p, x64, masks)
  cot_c, cot_f = torch.randn(oc.shape, generator=g).double(), torch.randn(of.shape, generator=g).double()
  ((oc * cot_c).sum() + (of * cot_f).sum()).backward()
  dev_p = {k: w[k].clone().requires_grad_(True) for k in train_encoder.PARAMS}
  xc, xf = train_encoder.encoder_forward(dev_p, imgs.to(device).permute(0, 3, 1, 2))
  assert ops._channels_last_view(xc) is not None and ops._channels_last_view(xf) is not None, 'training encoder outputs must be channels-last in memory'
  assert_close(xc, oc0, 1e-4, 1e-4, f'training encoder {name} coarse (|ref| up to {float(oc0.abs().max()):.1f})')
  assert_close(xf, of0, 1e-4, 1e-4, f'training encoder {name} fine')
  ((xc * cot_c.float().to(device)).sum() + (xf * cot_f.float().to(device)).sum()).backward()
  worst = 0.0
  for k in train_encoder.PARAMS:
    ref = ref_p[k].grad
    got = dev_p[k].grad
    assert got is not None, f'training encoder: no gradient for {k}'
    scale = float(ref.abs().max())
    # fp32-class products, fp32 sums in another order (atomics, split reductions): 3e-5 of the tensor's largest gradient + 1e-4 relative
    assert_close(got, ref, 3e-5 * scale + 1e-7, 1e-4, f'training encoder {name} grad {k} (max |g| {scale:.2e})')
    worst = max(worst, float((cpu(got).double() - ref).abs().max()) / max(scale, 1e-30))
  print(f'  training encoder {name}: {flips} ReLU decision(s) at arguments within rounding of zero differ from the oracle; worst gradient error {worst:.1e} of the tensor maximum')
  # the wrapper: training form under grad mode, forward-only kernels under no_grad, same maps
  class Holder(torch.nn.Module):
    def __init__(self):
      super().__init__()
      self.p = torch.nn.ParameterDict({k.replace('.', '__'): torch.nn.Parameter(v.detach().clone()) for k, v in dev_p.items()})

    def named_parameters(self, *a, **kw):
      return [(k.replace('__', '.'), v) for k, v in self.p.items()]

    def state_dict(self, *a, **kw):
      return {k.replace('__', '.'): v.detach() for k, v in self.p.items()}

  net = feature_network.ResNet.from_module(Holder())
  x = imgs.to(device).permute(0, 3, 1, 2)
  tc, tf = net(x)
  assert tc.requires_grad, 'wrapper under grad mode with trainable parameters must return maps with a graph'
  (tc.square().mean() + tf.mean()).backward()  # what the renderer's feature-map gradient does: into the wrapped module's parameters
  for k, v in net._source.named_parameters():
    assert v.grad is not None and bool(torch.isfinite(v.grad).all()) and float(v.grad.abs().max()) > 0.0, f'wrapper: no gradient reached {k}'
  with torch.no_grad():
    ic, _ = net(x)
  assert not ic.requires_grad
  assert_close(tc.detach(), ic, 5e-5, 1e-4, f'training encoder {name} vs forward-only kernels')
  return worst


def check_encoder_feeds_gather(device):
  """Maps produced by the HIP encoder are tapped in place (no repack): the gather on them equals the gather on an NCHW copy."""
  from dynibar_amd import feature_network
  scene, o, d, uv, _ = cases.scene_case('small')
  sd = to_dev(scene, device)
  enc = feature_network.ResNet.from_module(syn_encoder())
  _, fine = enc(sd['static_src_rgbs'][0].permute(0, 3, 1, 2))
  assert fine.shape[1] == 32
  pts_r, z_r, _ = O.sample_along_camera_ray(o, d, scene['depth_range'], 16, True, True)
  va = ops.SourceViews(sd['camera'], sd['static_src_rgbs'], sd['static_src_cameras'], fine)
  assert va.feat_cl.data_ptr() == fine.data_ptr(), 'channels-last maps must be used without a copy'
  vb = ops.SourceViews(sd['camera'], sd['static_src_rgbs'], sd['static_src_cameras'], fine.contiguous())
  assert vb.feat_cl.data_ptr() != fine.data_ptr()
  ra = ops.project_gather(va, o.shape[0], 16, ray_o=o.to(device), ray_d=d.to(device), z_vals=z_r.to(device))
  rb = ops.project_gather(vb, o.shape[0], 16, ray_o=o.to(device), ray_d=d.to(device), z_vals=z_r.to(device))
  assert_bitexact(ra[0], rb[0], 'gather on in-place channels-last maps')


def check_encoder_trains_through_gather(device):
  """The renderer's feature-map gradient reaches the encoder's parameters: maps from the ResNet wrapper under grad mode (training form) are
  tapped by the differentiable gather, and a loss on the gathered features fills the wrapped module's .grad with the same values as feeding
  the gather's map gradient to the encoder's backward by hand (two autograd Functions composed by torch)."""
  from dynibar_amd import feature_network, train_encoder, train_motion
  scene, o, d, uv, _ = cases.scene_case('small')
  sd = to_dev(scene, device)
  params = {k: torch.from_numpy(np.asarray(v)).float().to(device).requires_grad_(True) for k, v in syn_encoder().items() if k in train_encoder.PARAMS}
  imgs = sd['static_src_rgbs'][0]
  R, S = o.shape[0], 16
  pts_r, z_r, _ = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  kw = dict(ray_o=o.to(device), ray_d=d.to(device), z_vals=z_r.to(device))
  g = torch.Generator().manual_seed(9)

  def loss_of(fine):
    views = ops.SourceViews(sd['camera'], sd['static_src_rgbs'], sd['static_src_cameras'], fine.detach())
    rgb_feat = train_motion.gather(views, fine, R, S, **kw)[0]
    return rgb_feat

  _, fine = train_encoder.encoder_forward(params, imgs.permute(0, 3, 1, 2))
  rf = loss_of(fine)
  cot = torch.randn(rf.shape, generator=g).to(device)
  (rf * cot).sum().backward()
  got = {k: v.grad.clone() for k, v in params.items()}
  assert all(bool(torch.isfinite(v).all()) for v in got.values()) and float(got['conv1.weight'].abs().max()) > 0.0
  # by hand: the gather's gradient w.r.t. a leaf copy of the maps, pushed through the encoder separately
  for v in params.values():
    v.grad = None
  _, fine2 = train_encoder.encoder_forward(params, imgs.permute(0, 3, 1, 2))
  leaf = fine2.detach().clone().requires_grad_(True)
  (loss_of(leaf) * cot).sum().backward()
  fine2.backward(leaf.grad)
  for k, v in params.items():
    if k == 'out_conv.weight' or k == 'out_conv.bias':
      continue  # (rows of the coarse half get exact zeros in both)
    assert_close(v.grad, got[k], 2e-5 * float(got[k].abs().max()) + 1e-12, 1e-5, f'encoder through gather: grad {k}')


def syn_encoder():
  from dynibar_amd import synthetic
  return synthetic.make_encoder_weights(0)


def check_module_helpers(device, golden, name='small', S=64, with_fine=True):
  """The helper functions of ibrnet.render_ray that scripts may import directly (sample_pdf, compute_traj_pts, compute_optical_flow,
  compute_*_plucker_coordinate, fine_render_rays), with the reference's signatures, against the real reference's outputs."""
  import types
  from dynibar_amd import projection, render_ray as RR
  scene, o, d, uv, _ = cases.scene_case(name)
  dv = lambda x: x.to(device)
  # Pluecker coordinates
  pts_r, z_r, _ = O.sample_along_camera_ray(o, d, scene['depth_range'], S, True, True)
  assert_close(RR.compute_ref_plucker_coordinate(dv(o), dv(d)), torch.from_numpy(golden['plucker/ref']), 1e-6, 1e-6, f'{name} ref Pluecker')
  assert_close(RR.compute_src_plucker_coordinate(dv(pts_r), dv(scene['static_src_cameras'])), torch.from_numpy(golden['plucker/src']), 2e-6, 1e-6,
               f'{name} src Pluecker')
  # sample_pdf on the reference's own composite weights, both parametrisations; weights are modified in place like the reference's
  z = torch.from_numpy(golden['composite/z_vals']); w = torch.from_numpy(golden['composite/weights'])
  for inv in (True, False):
    if inv:
      iz = 1.0 / z
      bins = torch.flip(0.5 * (iz[:, 1:] + iz[:, :-1]), dims=[1]); ww = torch.flip(w[:, 1:-1], dims=[1]).contiguous()
    else:
      bins = 0.5 * (z[:, 1:] + z[:, :-1]); ww = w[:, 1:-1].contiguous()
    wd = dv(ww.clone())
    smp = RR.sample_pdf(dv(bins.contiguous()), wd, S, det=True)
    assert_bitexact(wd, ww + 1e-5, 'sample_pdf adds 1e-5 to its weights argument in place')
    tie, tol = O.cdf_sample_conditioning(z, w, S, inv, True, None)
    ref = torch.from_numpy(golden[f'pdf/inv{int(inv)}/det'])
    over = ((cpu(smp) - ref).abs() > tol + 2e-6 * ref.abs() + 1e-7) & ~tie
    assert int(over.sum()) == 0, f'sample_pdf inv={inv}: {int(over.sum())} samples beyond the conditioning bound'
  # compute_traj_pts and compute_optical_flow on the reference's motion coefficients / trajectory points
  coeff = torch.from_numpy(golden['motion/coeff_raw']).clone()
  B = cases.NUM_BASIS
  basis = O.init_dct_basis(B, cases.NUM_FRAMES)
  row = basis[None, None, cases.REF_FRAME + 2, :]
  got = RR.compute_traj_pts(dv(coeff[..., :B]), dv(coeff[..., B:2 * B]), dv(coeff[..., 2 * B:]), dv(row))
  assert_close(got, O.compute_traj_pts(coeff[..., :B], coeff[..., B:2 * B], coeff[..., 2 * B:], row), 1e-6, 1e-5, f'{name} compute_traj_pts')
  pts_seq = torch.from_numpy(golden['motion/pts_seq'])
  flows = RR.compute_optical_flow({'weights': dv(w)}, dv(pts_seq), dv(scene['src_cameras']), dv(uv))
  assert_close(flows, torch.from_numpy(golden['flow/render_flows']), 2e-4, 1e-5, f'{name} compute_optical_flow')
  if not with_fine:
    return
  # fine_render_rays on explicit networks = the fine pass of render_rays_mv, bit for bit (same kernels, same inputs)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  proj = projection.Projector(device)
  batch = make_ray_batch(scene, o, d, uv, device)
  cfeat = (dv(scene['featmaps']), None, dv(scene['static_featmaps']))
  ffeat = (dv(scene['featmaps_fine']), None, dv(scene['static_featmaps_fine']))
  ret = RR.render_rays_mv((fidx, None), (dv(temb), None), (toff, None), batch, model, proj, cfeat, ffeat, S, args, inv_uniform=True, N_importance=S,
                          det=True, is_train=False)
  z_all = ret['outputs_fine_ref']['z_vals']
  pts_f, s_all = ops.points_from_z(batch['ray_o'], batch['ray_d'], z_all, batch['depth_range'])
  out, out_dy, a, b = RR.fine_render_rays(proj, batch, ffeat, pts_f, z_all, s_all, dv(temb), None, fidx, None, toff, None, model.net_fine_dy, model.net_fine_st,
                                          model.motion_mlp_fine, model.trajectory_basis_fine, 0, False)
  assert a is None and b is None and list(out.keys()) == list(ret['outputs_fine_ref'].keys())
  for k in out:
    assert_bitexact(out[k], ret['outputs_fine_ref'][k], f'fine_render_rays {k}')
  for k in out_dy:
    assert_bitexact(out_dy[k], ret['outputs_fine_ref_dy'][k], f'fine_render_rays dy {k}')


def sampler_data():
  """The seeded `data` dict of tests/golden/make_golden.py:sampler_goldens."""
  scene, o, d, uv, pix = cases.scene_case('small')
  H, W = 48, 64
  rs = np.random.RandomState(0)
  f32 = lambda a: torch.from_numpy(a.astype(np.float32))
  return dict(camera=scene['camera'], rgb_path='x', depth_range=scene['depth_range'], src_rgbs=scene['src_rgbs'], src_cameras=scene['src_cameras'],
              static_src_rgbs=scene['static_src_rgbs'], static_src_cameras=scene['static_src_cameras'], rgb=f32(rs.rand(1, H, W, 3)),
              disp=f32(rs.rand(1, H, W)), motion_mask=f32(rs.rand(1, H, W) > 0.5), static_mask=f32(rs.rand(1, H, W) > 0.5),
              flows=f32(rs.rand(1, 6, H, W, 2)), masks=f32(rs.rand(1, 6, H, W)), anchor_camera=scene['camera'])


def check_ray_sampler(device, golden):
  from dynibar_amd import sample_ray as SR
  data = sampler_data()
  SR.rng = np.random.RandomState(234)
  smp = SR.RaySamplerSingleImage(data, device)
  assert_close(smp.rays_o, torch.from_numpy(golden['rays_o']), 0.0, 0.0, 'rays_o')
  assert_close(smp.rays_d, torch.from_numpy(golden['rays_d']), 1e-6, 2e-6, 'rays_d')
  assert_bitexact(smp.uv_grid, torch.from_numpy(golden['uv_grid']), 'uv_grid')
  for i, mode in enumerate(['uniform', 'center', 'uniform']):
    rb = smp.random_sample(37, mode, 0.8)
    assert np.array_equal(np.asarray(rb['selected_inds']), golden[f'rand{i}/selected_inds']), f'selected_inds draw {i} differs from the reference'
    assert_close(rb['ray_d'], torch.from_numpy(golden[f'rand{i}/ray_d']), 1e-6, 2e-6, f'rand{i} ray_d')
    assert_bitexact(rb['rgb'], torch.from_numpy(golden[f'rand{i}/rgb']), f'rand{i} rgb')
    assert_bitexact(rb['flows'], torch.from_numpy(golden[f'rand{i}/flows']), f'rand{i} flows')
  smp2 = SR.RaySamplerSingleImage(data, device, render_stride=2)
  assert_close(smp2.rays_d, torch.from_numpy(golden['stride2/rays_d']), 1e-6, 2e-6, 'stride-2 rays_d')
  allb = smp.get_all()
  assert allb['ray_o'].shape == (48 * 64, 3) and allb['flows'].shape == (6, 48 * 64, 2) and allb['disp'].shape == (48 * 64,)


def image_case(device):
  from dynibar_amd import synthetic as syn
  cfg = dict(seed=4, H=12, W=16, V=7, n_static=8, smooth=True)
  sc = syn.make_scene(**cfg)
  fine = syn.make_scene(**dict(cfg, tag=1))
  scene = {k: cases.t(v) for k, v in sc.items()}
  data = dict(camera=scene['camera'], rgb_path='x', depth_range=scene['depth_range'], src_rgbs=scene['src_rgbs'], src_cameras=scene['src_cameras'],
              static_src_rgbs=scene['static_src_rgbs'], static_src_cameras=scene['static_src_cameras'])
  cfeat = (scene['featmaps'].to(device), None, scene['static_featmaps'].to(device))
  ffeat = (cases.t(fine['featmaps']).to(device), None, cases.t(fine['static_featmaps']).to(device))
  return data, cfeat, ffeat


def check_render_image_nvi(device, golden, chunk_size=80):
  """render_single_image_nvi on a 12x16 frame in 3 chunks against the real reference's frame."""
  import types
  from dynibar_amd import projection, render_image, sample_ray
  data, cfeat, ffeat = image_case(device)
  smp = sample_ray.RaySamplerSingleImage(data, device)
  rb = smp.get_all()
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  fidx, temb, toff = cases.time_args(7)
  ret = render_image.render_single_image_nvi((fidx, None), (temb.to(device), None), (toff, None), smp, rb, model, projection.Projector(device),
                                             chunk_size, 64, args, inv_uniform=True, N_importance=64, det=True, coarse_featmaps=cfeat,
                                             fine_featmaps=ffeat, is_train=False)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_fine_ref'):
    for k, v in ret[grp].items():
      ref = torch.from_numpy(golden[f'{grp}/{k}'])
      assert tuple(v.shape) == tuple(ref.shape), f'{grp}/{k}: shape {tuple(v.shape)} vs reference {tuple(ref.shape)}'
      assert v.device.type == 'cpu', 'render_single_image_* returns host tensors like the reference'
      if ref.dtype == torch.bool:
        assert_bitexact(v, ref, f'{grp}/{k}')
      else:
        tol = _group_tol(k, 'small')
        assert_close(v.float(), ref.float(), tol['atol'], tol['rtol'], f'{grp}/{k}')
      n += 1
  assert n == len(golden), 'render_single_image_nvi output key set differs from the reference'
  assert ret['outputs_fine'] is None
  return ret


def check_chunk_stream_invariance(device, chunk_size=40):
  """render_single_image_nvi with its chunks on 1, 2 and 3 alternating HIP streams (render_image.CHUNK_STREAMS; per-stream network workspaces): identical
  bits in every entry.  (Round 5 found the failure mode this guards: two chunks in flight sharing one network workspace.)"""
  import types
  from dynibar_amd import projection, render_image, sample_ray
  data, cfeat, ffeat = image_case(device)
  smp = sample_ray.RaySamplerSingleImage(data, device)
  rb = smp.get_all()
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0, frame_outputs='all')
  fidx, temb, toff = cases.time_args(7)
  outs = []
  prev = render_image.CHUNK_STREAMS
  try:
    for n in (1, 2, 3, 2):
      render_image.CHUNK_STREAMS = n
      ret = render_image.render_single_image_nvi((fidx, None), (temb.to(device), None), (toff, None), smp, rb, model, projection.Projector(device),
                                                 chunk_size, 64, args, inv_uniform=True, N_importance=64, det=True, coarse_featmaps=cfeat,
                                                 fine_featmaps=ffeat, is_train=False)
      outs.append({(g, k): v.clone() if isinstance(v, torch.Tensor) else v for g in ('outputs_coarse_ref', 'outputs_fine_ref') for k, v in ret[g].items()})
  finally:
    render_image.CHUNK_STREAMS = prev
  n = 0
  for other in outs[1:]:
    for key, v in outs[0].items():
      if isinstance(v, torch.Tensor):
        assert_bitexact(other[key], v, f'chunk streams: {key}')
        n += 1
  return n


def check_chunk_stream_invariance_full_size(device, frames=2):
  """The SAME invariance at the size bench.py times a frame at (BASELINE configs[2]: 288 x 512 rays, 64 + 64 samples, 7 + 11 views, chunk 8192) -- where the
  chunks of two streams really overlap on the device.  Round 6 found the small-chunk check above blind to it: with two streams 50-100 rays of a full frame
  differed by up to 1e-3 from run to run (k_static_ref_feat's waves beside a one-wave-per-SIMD kernel: csrc/dyn_mlp.h, DYN_EXCLUSIVE_CU).  Every ray of
  `frames` two-stream frames (the first one cold: per-view caches and network packing inside the frame) must equal the one-stream frame bit for bit."""
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  from dynibar_amd import render_image
  from frame_case import FrameCase
  prev = render_image.CHUNK_STREAMS
  keys = (('outputs_coarse_ref', 'rgb'), ('outputs_fine_ref', 'rgb'), ('outputs_fine_ref', 'depth'))
  try:
    outs = []
    for n in [2] * frames + [1]:
      render_image.CHUNK_STREAMS = n
      fc = FrameCase(device) if not outs or n == 1 else fc  # (a fresh case for the first two-stream frame and for the one-stream frame: both start cold)
      smp, rb = fc.sampler()
      ret = fc.render(smp, rb)
      torch.cuda.synchronize()
      outs.append({k: ret[k[0]][k[1]].clone() for k in keys})
  finally:
    render_image.CHUNK_STREAMS = prev
  n = 0
  for i, o in enumerate(outs[:-1]):
    for k in keys:
      assert_bitexact(o[k], outs[-1][k], f'full frame, two chunk streams (frame {i}) vs one stream: {k[0]}/{k[1]}')
      n += 1
  return n


def check_checkpoint_render_chain(device, g_img, g_enc, tmpdir, chunk_size=80):
  """Section 8f-4 on the device: both checkpoint files of the reference (model.py:424-441 coarse / monocular, :177-190 fine) are WRITTEN with torch.save
  from nn.Module state dicts (de-parallelised like model.py:13-15; the encoders with the decoder layers forward never runs riding along, optimizer /
  scheduler entries present), READ with checkpoint.load_model, and the loaded object is used as it is: its HIP encoders on the seeded images of the
  encoder golden, and render_single_image_nvi with the loaded nets / bases against the real reference's frame (tests/golden/image_nvi.npz)."""
  import types
  import refmodules
  from dynibar_amd import checkpoint, projection, render_image, sample_ray
  W = cases.model_weights(0)
  margs = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False)
  kind_of = lambda k: 'static' if k.endswith('_st') else ('dynamic' if '_dy' in k else 'motion')
  mods = {k: torch.nn.DataParallel(refmodules.load_numpy_state(refmodules.like_reference(kind_of(k), margs), v)) for k, v in W.items()}
  de_parallel = lambda m: m.module if hasattr(m, 'module') else m  # model.py:13-15
  imgs, enc_sd = cases.encoder_case('small')
  g = torch.Generator().manual_seed(3)

  def encoder_file_dict(sd):
    d = {k: torch.from_numpy(v) for k, v in sd.items()}
    # the reference's ResNet also owns layer2 / layer3 / upconv* / iconv* (feature_network.py:232-245): saved with the rest, never executed
    for name, shape in (('layer2.0.conv1.weight', (128, 64, 3, 3)), ('layer3.0.conv1.weight', (256, 128, 3, 3)), ('upconv3.conv.conv.weight', (128, 256, 3, 3)),
                        ('upconv3.conv.bn.weight', (128,)), ('iconv3.conv.weight', (128, 256, 3, 3)), ('upconv2.conv.conv.weight', (64, 128, 3, 3)),
                        ('iconv2.conv.weight', (64, 128, 3, 3)), ('iconv2.bn.bias', (64,))):
      d[name] = torch.randn(shape, generator=g)
    return d

  basis = torch.nn.Parameter(O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES))
  coarse = {'optimizer': {'state': {}, 'param_groups': [{'lr': 5e-4}]}, 'scheduler': {'last_epoch': 7}, 'net_coarse_st': de_parallel(mods['net_coarse_st']).state_dict(),
            'net_coarse_dy': de_parallel(mods['net_coarse_dy']).state_dict(), 'feature_net': encoder_file_dict(enc_sd), 'motion_mlp': de_parallel(mods['motion_mlp']).state_dict(),
            'traj_basis': basis, 'global_step': 250000}
  fine = {'optimizer': {}, 'scheduler': {}, 'net_fine_st': de_parallel(mods['net_fine_st']).state_dict(), 'net_fine_dy': de_parallel(mods['net_fine_dy']).state_dict(),
          'feature_net_fine': encoder_file_dict(cases.encoder_case('odd')[1]), 'motion_mlp_fine': de_parallel(mods['motion_mlp_fine']).state_dict(),
          'traj_basis_fine': basis.detach().clone(), 'global_step': 60000}
  pc, pf = os.path.join(str(tmpdir), 'model_250000.pth'), os.path.join(str(tmpdir), 'model_fine_060000.pth')
  torch.save(coarse, pc)
  torch.save(fine, pf)
  model = checkpoint.load_model(pc, pf, device=device)
  assert model.global_step == 60000 and model.trajectory_basis.device.type == 'cuda' and not model.trajectory_basis.requires_grad
  assert getattr(model.net_coarse_dy, 'shift', None) == 0.0, 'an Nvidia-benchmark checkpoint (no feature_net_st): DynibarFF builds the dynamic net with shift 0'
  # ---- the loaded HIP encoder against the real reference's encoder outputs ----
  xc, xf = model.feature_net(imgs.to(device).permute(0, 3, 1, 2))
  for got, key in ((xc, 'coarse'), (xf, 'fine')):
    assert_close(got, torch.from_numpy(g_enc[f'small/{key}']), 1e-4, 1e-4, f'checkpoint -> encoder small {key}')
  # ---- the loaded nets / bases render the reference's frame ----
  data, cfeat, ffeat = image_case(device)
  smp = sample_ray.RaySamplerSingleImage(data, device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  fidx, temb, toff = cases.time_args(7)
  ret = render_image.render_single_image_nvi((fidx, None), (temb.to(device), None), (toff, None), smp, smp.get_all(), model, projection.Projector(device), chunk_size, 64,
                                             args, inv_uniform=True, N_importance=64, det=True, coarse_featmaps=cfeat, fine_featmaps=ffeat, is_train=False)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_fine_ref'):
    for k in ('rgb', 'depth', 'mask', 'weights'):
      if f'{grp}/{k}' not in g_img:
        continue
      v, ref = ret[grp][k], torch.from_numpy(g_img[f'{grp}/{k}'])
      if ref.dtype == torch.bool:
        assert_bitexact(v, ref, f'checkpoint -> frame {grp}/{k}')
      else:
        tol = _group_tol(k, 'small')
        assert_close(v.float(), ref.float(), tol['atol'], tol['rtol'], f'checkpoint -> frame {grp}/{k}')
      n += 1
  assert n >= 6
  return n


def check_render_image_mono(device, golden, chunk_size=80):
  """render_single_image_mono on a 12x16 frame in 3 chunks (5 time-offset views + 2 virtual views) against the real reference's frame."""
  import types
  from dynibar_amd import projection, render_image, sample_ray
  data, cfeat, _ = image_case(device)
  smp = sample_ray.RaySamplerSingleImage(data, device)
  rb = smp.get_all()
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  fidx, temb, toff = cases.time_args(7)
  ret = render_image.render_single_image_mono((fidx, None), (temb.to(device), None), (toff[:5], None), smp, rb, model, projection.Projector(device),
                                              chunk_size, 64, args, inv_uniform=True, N_importance=0, det=True, featmaps=cfeat, is_train=False,
                                              num_vv=2)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_coarse_st', 'outputs_coarse_anchor'):
    keys_ref = sorted(k[len(grp) + 1:] for k in golden if k.startswith(grp + '/'))
    assert sorted(ret[grp].keys()) == keys_ref, f'{grp}: keys {sorted(ret[grp].keys())} vs reference {keys_ref}'
    for k, v in ret[grp].items():
      ref = torch.from_numpy(golden[f'{grp}/{k}'])
      assert tuple(v.shape) == tuple(ref.shape), f'{grp}/{k}: shape {tuple(v.shape)} vs reference {tuple(ref.shape)}'
      assert v.device.type == 'cpu', 'render_single_image_* returns host tensors like the reference'
      if ref.dtype == torch.bool:
        assert_bitexact(v, ref, f'{grp}/{k}')
      else:
        tol = _group_tol(k, 'small')
        assert_close(v.float(), ref.float(), tol['atol'], tol['rtol'], f'{grp}/{k}')
      n += 1
  assert n == len(golden), 'render_single_image_mono output key set differs from the reference'
  assert ret['outputs_fine'] is None
  return ret


def check_render_image_mono_train(device, golden, chunk_size=80):
  """render_single_image_mono(is_train=True): the anchor group is assembled too; 4-D entries stay per-chunk lists like the reference's."""
  import types
  from dynibar_amd import projection, render_image, sample_ray, synthetic as syn
  sc = syn.make_scene(seed=4, H=12, W=16, V=7, n_static=8, smooth=True)
  scene = {k: cases.t(v) for k, v in sc.items()}
  scn, fi, te, to = cases.anchor_case(scene, 2, 1)
  data = dict(camera=scn['camera'], rgb_path='x', depth_range=scn['depth_range'], src_rgbs=scn['src_rgbs'], src_cameras=scn['src_cameras'],
              static_src_rgbs=scn['static_src_rgbs'], static_src_cameras=scn['static_src_cameras'],
              anchor_src_rgbs=scn['anchor_src_rgbs'], anchor_src_cameras=scn['anchor_src_cameras'])
  smp = sample_ray.RaySamplerSingleImage(data, device)
  rb = smp.get_all()
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  feat = (scn['featmaps'].to(device), scn['featmaps_anchor'].to(device), scn['static_featmaps'].to(device))
  ret = render_image.render_single_image_mono(fi, (te[0].to(device), te[1].to(device)), to, smp, rb, model, projection.Projector(device), chunk_size,
                                              64, args, inv_uniform=True, N_importance=0, det=True, featmaps=feat, is_train=True, num_vv=2)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_coarse_st', 'outputs_coarse_anchor'):
    keys_ref = sorted({k[len(grp) + 1:].split('#')[0] for k in golden if k.startswith(grp + '/')})
    assert sorted(ret[grp].keys()) == keys_ref, f'{grp}: keys {sorted(ret[grp].keys())} vs reference {keys_ref}'
    for k, v in ret[grp].items():
      parts = v if isinstance(v, list) else [v]
      names = [f'{grp}/{k}#{i}' for i in range(len(parts))] if isinstance(v, list) else [f'{grp}/{k}']
      assert all(nm in golden for nm in names) and (not isinstance(v, list) or f'{grp}/{k}#{len(parts)}' not in golden), f'{grp}/{k}: chunk list length'
      for nm, t in zip(names, parts):
        ref = torch.from_numpy(golden[nm])
        assert tuple(t.shape) == tuple(ref.shape), f'{nm}: shape {tuple(t.shape)} vs reference {tuple(ref.shape)}'
        assert t.device.type == 'cpu'
        if ref.dtype == torch.bool:
          assert_bitexact(t, ref, nm)
        else:
          tol = _group_tol(k, 'small')
          assert_close(t.float(), ref.float(), tol['atol'], tol['rtol'], nm)
        n += 1
  assert n == len(golden), 'render_single_image_mono(is_train=True) output set differs from the reference'
  return n


def check_full_size_properties(device, R=4096, S=64, V=8, N_importance=64):
  """BASELINE configs[1] at its full size (4096 rays x 64 samples x 8 views, 288x512 sources), where the oracle would take minutes:
  size-independent properties + an oracle spot check on a few rays."""
  from dynibar_amd import synthetic as syn
  sc = syn.make_scene(seed=0, H=288, W=512, V=V, n_static=V)
  scene = {k: torch.from_numpy(v).to(device) for k, v in sc.items()}
  pix = syn.sample_pixels(7, 288, 512, R)
  o_np, d_np, _ = syn.pixel_rays(sc['camera'], pix)
  o, d = torch.from_numpy(o_np).to(device), torch.from_numpy(d_np).to(device)
  wts = syn.make_weights('static', 0)
  net = ops.StaticNet(wts, device, True, False)
  out, raw = run_static_pass(device, scene, net, o, d, S)
  # (1) chunk invariance: rays are independent, so rendering the batch in two halves is bit-identical (what multi-GPU tiling relies on)
  h = R // 2
  out_a, _ = run_static_pass(device, scene, net, o[:h], d[:h], S)
  out_b, _ = run_static_pass(device, scene, net, o[h:], d[h:], S)
  for k in ('rgb', 'depth', 'weights', 'mask'):
    assert_bitexact(torch.cat([out_a[k], out_b[k]], 0), out[k], f'chunk invariance of {k}')
  # (2) compositing invariants (render_ray.py:185-199): alpha, weights in [0,1], sum of weights <= 1, depth inside the sampled range
  w = cpu(out['weights'])
  assert float(w.min()) >= 0.0 and float(w.max()) <= 1.0 + 1e-6 and float(w.sum(dim=1).max()) <= 1.0 + 1e-5
  z = cpu(out['z_vals'])
  dep = cpu(out['depth'])
  assert bool((dep <= z[:, -1] * (1 + 1e-5)).all()) and bool((dep >= 0).all())
  assert bool(torch.isfinite(cpu(raw)[..., :3]).all()) and float(cpu(out['rgb']).min()) >= -1e-6 and float(cpu(out['rgb']).max()) <= 1.0 + 1e-5
  # (3) importance resampling at full size: sorted, inside [near, far], the coarse depths are a subset (bitwise)
  z_all, z_s, _ = ops.fine_samples(out['z_vals'], out['weights'], N_importance, True)
  za = cpu(z_all)
  near, far = float(sc['depth_range'][0, 0]), float(sc['depth_range'][0, 1])
  assert bool((za[:, 1:] >= za[:, :-1]).all()) and float(za.min()) >= near * (1 - 1e-6) and float(za.max()) <= far * (1 + 1e-6)
  merged = torch.sort(torch.cat([z, cpu(z_s)], 1), 1)[0]
  assert_bitexact(za, merged, 'fine depths = sorted union of coarse and new depths')
  # (4) oracle spot check on 48 of the 4096 rays
  idx = torch.arange(0, R, R // 48)[:48]
  cs = {k: torch.from_numpy(v) for k, v in sc.items()}
  ref = O.static_branch_pass(O.tdict(wts), cs, torch.from_numpy(o_np)[idx], torch.from_numpy(d_np)[idx], S, True, True)
  assert_close(cpu(out['rgb'])[idx], ref['rgb'], 1e-4, 0.0, 'full-size rgb vs oracle (48 rays)')
  assert_close(cpu(out['depth'])[idx], ref['depth'], 0.0, 3e-4, 'full-size depth vs oracle (48 rays)')
  return float((cpu(out['rgb'])[idx] - ref['rgb']).abs().max())


def _full_frame_subset(ret_group, idx, HW):
  """pixel entries of a full-frame group ([H,W,...] host tensors) on the strided ray subset idx"""
  out = {}
  for k in ('rgb', 'depth', 'mask'):
    if k in ret_group and ret_group[k] is not None:
      v = ret_group[k]
      out[k] = v.reshape((HW,) + tuple(v.shape[2:]))[idx]
  return out


def check_full_frames_vs_oracle(device, stride=563, which=('nvi', 'mono')):
  """BASELINE configs[2] and configs[3] AT THEIR FULL SIZE (288 x 512 rays, chunk 8192 -- the frames bench.py times): render_single_image_nvi
  (64 + 64 samples, 7 dynamic + 11 static views) and render_single_image_mono (kid-running arguments: 64 samples, 7 + 3 dynamic and 15 static views,
  anti_alias_pooling 0, mask_rgb 1) on the HIP path, compared with the oracle on a strided subset of the frame's rays (every `stride`-th ray:
  262 rays at the default; the oracle needs seconds for them, hours for the frame).  Together with the chunk-invariance property (bit-exact,
  check_full_size_properties) this holds the frames that are timed to the reference's arithmetic, not only their 12 x 16 twins."""
  import types
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  from dynibar_amd import synthetic as syn
  worst = {}
  W = oracle_models()  # the weights FrameCase / MonoFrame use: syn.make_weights seeds 0 (coarse) and 100 (fine), DCT basis 6 x 24
  if 'nvi' in which:
    from frame_case import FrameCase
    fc = FrameCase(device)
    smp, rb = fc.sampler()
    ret = fc.render(smp, rb)
    HW = fc.H * fc.W
    idx = torch.arange(0, HW, stride)
    sc = syn.make_scene(seed=0, H=fc.H, W=fc.W, V=fc.vdy, n_static=fc.vst)
    fine = syn.make_scene(seed=0, H=fc.H, W=fc.W, V=fc.vdy, n_static=fc.vst, tag=1)
    scene = {k: cases.t(v) for k, v in sc.items()}
    scene['featmaps_fine'], scene['static_featmaps_fine'] = cases.t(fine['featmaps']), cases.t(fine['static_featmaps'])
    o, d, uv = (cpu(rb[k])[idx] for k in ('ray_o', 'ray_d', 'uv_grid'))
    run = lambda: O.render_rays_mv(W, dict(scene), o, d, uv, fc.fidx, torch.tensor([fc.fidx / 24.0]), fc.toff, 64, 64)
    ref = run()
    sens = projection_sensitivity(run)  # the matrices are formed on the device (rocSOLVER) here, by LAPACK in the oracle: the reference's own conditioning
    for grp in ('outputs_coarse_ref', 'outputs_fine_ref'):
      got = _full_frame_subset(ret[grp], idx, HW)
      for k, v in got.items():
        r = ref[grp][k]
        if r.dtype == torch.bool:
          assert_bitexact(v, r, f'full nvi frame {grp}/{k} ({len(idx)} of {HW} rays)')
        else:
          tol = _group_tol(k)
          ex = sens[grp].get(k)
          assert_close(v.float(), r.float(), tol['atol'], tol['rtol'], f'full nvi frame {grp}/{k} ({len(idx)} of {HW} rays)', extra=None if ex is None else SENS_FACTOR * ex)
          worst[f'nvi/{grp}/{k}'] = float((v.float() - r.float()).abs().max())
    del fc, ret, smp, rb
  if 'mono' in which:
    from config_cases import MonoFrame
    mf = MonoFrame(device)
    ret = mf.render()
    H, Wd = 288, 512
    HW = H * Wd
    idx = torch.arange(0, HW, stride)
    sc = syn.make_scene(seed=31, H=H, W=Wd, V=10, n_static=15)
    scene = {k: cases.t(v) for k, v in sc.items()}
    from dynibar_amd import sample_ray
    rb = sample_ray.RaySamplerSingleImage(mf.data, device).get_all()
    o, d, uv = (cpu(rb[k])[idx] for k in ('ray_o', 'ray_d', 'uv_grid'))
    run = lambda: O.render_rays_mono_eval(W, dict(scene), o, d, uv, mf.fidx, torch.tensor([mf.fidx / 24.0]), mf.toff, 64, anti_alias_pooling=False, mask_rgb=True,
                                          num_vv=mf.num_vv)
    ref = run()
    sens = projection_sensitivity(run)
    for grp in ('outputs_coarse_ref', 'outputs_coarse_st'):
      got = _full_frame_subset(ret[grp], idx, HW)
      for k, v in got.items():
        r = ref[grp][k]
        if r.dtype == torch.bool:
          assert_bitexact(v, r, f'full mono frame {grp}/{k} ({len(idx)} of {HW} rays)')
        else:
          tol = _group_tol(k)
          ex = sens[grp].get(k)
          assert_close(v.float(), r.float(), tol['atol'], tol['rtol'], f'full mono frame {grp}/{k} ({len(idx)} of {HW} rays)', extra=None if ex is None else SENS_FACTOR * ex)
          worst[f'mono/{grp}/{k}'] = float((v.float() - r.float()).abs().max())
  return worst


def check_render_rays_mono_vv(device, name='small', S=64, num_vv=2):
  import types
  from dynibar_amd import projection, render_ray
  scene, o, d, uv, _ = cases.scene_case(name)
  Vd = scene['src_rgbs'].shape[1]
  fidx, temb, toff = cases.time_args(Vd)
  toff = toff[:Vd - num_vv]  # the last num_vv source views are the virtual ones: no time offset, no displacement
  W = {k: O.tdict(v) for k, v in cases.model_weights(0).items()}
  W['trajectory_basis'] = O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  ref = O.render_rays_mono_eval(W, dict(scene), o, d, uv, fidx, temb, toff, S, True, True, num_vv=num_vv)
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
  batch = make_ray_batch(scene, o, d, uv, device)
  feat = (scene['featmaps'].to(device), None, scene['static_featmaps'].to(device))
  ret = render_ray.render_rays_mono((fidx, None), (temb.to(device), None), (toff, None), batch, model, feat, projection.Projector(device), S, args,
                                    inv_uniform=True, det=True, is_train=False, num_vv=num_vv)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_coarse_ref_dy', 'outputs_coarse_st'):
    assert list(ret[grp].keys()) == list(ref[grp].keys()), f'{grp}: key order differs from the reference'
    for k, v in ret[grp].items():
      r = ref[grp][k]
      if r.dtype == torch.bool:
        assert_bitexact(v, r, f'{grp}/{k}')
      else:
        tol = _group_tol(k, name)
        assert_close(cpu(v).float(), r.float(), tol['atol'], tol['rtol'], f'mono vv {grp}/{k}')
      n += 1
  return n


def check_render_rays_mono_train(device, golden, tag, shift, mode, name='small', S=64):
  """render_rays_mono(is_train=True): all five output groups (reference-time pass + cross-time rendering at the anchor frame)
  against the real reference's forward values."""
  import types
  from dynibar_amd import projection, render_ray
  scene, o, d, uv, _ = cases.scene_case(name)
  sc, fi, te, to = cases.anchor_case(scene, 2, shift)
  model = make_model(device)
  args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=mode)
  batch = make_ray_batch(sc, o, d, uv, device)
  batch['anchor_src_rgbs'] = sc['anchor_src_rgbs'].to(device)
  batch['anchor_src_cameras'] = sc['anchor_src_cameras'].to(device)
  feat = (sc['featmaps'].to(device), sc['featmaps_anchor'].to(device), sc['static_featmaps'].to(device))
  ret = render_ray.render_rays_mono(fi, (te[0].to(device), te[1].to(device)), to, batch, model, feat, projection.Projector(device), S, args,
                                    inv_uniform=True, det=True, is_train=True, num_vv=2)
  n = 0
  for grp in ('outputs_coarse_ref', 'outputs_coarse_ref_dy', 'outputs_coarse_st', 'outputs_coarse_anchor', 'outputs_coarse_anchor_dy'):
    keys_ref = [k[len(f'{tag}/{grp}/'):] for k in golden if k.startswith(f'{tag}/{grp}/')]
    assert sorted(ret[grp].keys()) == sorted(keys_ref), f'{grp}: keys {sorted(ret[grp].keys())} vs reference {sorted(keys_ref)}'
    n += check_group_vs_golden(f'{tag}/{grp}/', ret[grp], golden, name)
  return n



# ----------------------------------------------------------------------------------------------------------------------
# training of the static branch (first slice): values and gradients vs autograd through the oracle
# ----------------------------------------------------------------------------------------------------------------------
def _oracle_static_graph(sd, sc, o, d, S, aa, mask_rgb, exp_jitter=None):
  """static_branch_pass (the composition behind ret['outputs_coarse_st']) with the oracle's exp-jitter hook exposed"""
  pts, z, _ = O.sample_along_camera_ray(o, d, sc['depth_range'], S, True, True)
  Vs = sc['static_src_rgbs'].shape[1]
  rgb_feat, ray_diff, mask = O.compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), sc['camera'], sc['static_src_rgbs'],
                                                    sc['static_src_cameras'], sc['static_featmaps'])
  pm = mask[..., 0].sum(dim=2) > 1
  raw = O.static_net(sd, pts, O.ref_plucker(o, d), O.src_plucker(pts, sc['static_src_cameras']), rgb_feat, F.normalize(d, dim=-1), ray_diff, mask,
                     aa, mask_rgb, exp_jitter=exp_jitter)
  out = O.raw2outputs_vanilla(raw, z, pm)
  out['raw'] = raw
  return out, pts, mask


def train_static_reference(name, S, R, aa, mask_rgb, weights, seed=0, dtype=torch.float32, jitter_seed=None):
  """Oracle side: the static bootstrap graph (train.py:116-199) on torch-CPU autograd.  -> inputs, outputs, cotangents, gradients.
  dtype=float64 evaluates the same graph in double (how far the reference's own fp32 round-off moves a value or a gradient);
  jitter_seed perturbs exp() of the anti-alias pooling by +-1 ulp (see check_static_net)."""
  scene, o, d, uv, _ = cases.scene_case(name)
  if R is not None:
    o, d = o[:R], d[:R]
  sd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in O.tdict(_weights(weights)['net_coarse_st']).items() if (aa or k != 's')}
  sc = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in scene.items()}
  fm = scene['static_featmaps'].clone().to(dtype).requires_grad_(True)
  sc['static_featmaps'] = fm
  o, d = o.to(dtype), d.to(dtype)
  Vs = scene['static_src_rgbs'].shape[1]
  jit = None
  if jitter_seed is not None:
    jit = ((torch.randint(0, 3, (o.shape[0], S, Vs, 1), generator=torch.Generator().manual_seed(50 + jitter_seed)).float() - 1.0) * 6e-8).to(dtype)
  prev = torch.get_default_dtype()
  torch.set_default_dtype(dtype)  # the oracle's linspace / ones / tensor constructors follow the default dtype
  try:
    out, pts, _ = _oracle_static_graph(sd, sc, o, d, S, aa, mask_rgb, jit)
  finally:
    torch.set_default_dtype(prev)
  # rays with a sample on a frustum boundary may flip a mask bit between implementations (see check_static_pass): no cotangent for them
  keep = ~boundary_margin(pts.detach()[None].repeat(Vs, 1, 1, 1).float(), scene['static_src_cameras'][0]).any(dim=2).any(dim=1)
  g = torch.Generator().manual_seed(100 + seed)
  n = o.shape[0]
  cot = dict(rgb=torch.randn(n, 3, generator=g) * keep[:, None], depth=0.1 * torch.randn(n, generator=g) * keep,
             weights=0.3 * torch.randn(n, S, generator=g) * keep[:, None])
  loss = sum((out[k] * cot[k].to(dtype)).sum() for k in cot)
  loss.backward()
  grads = {k: v.grad.detach() for k, v in sd.items()}
  grads['featmaps'] = fm.grad.detach()
  vals = {k: out[k].detach() for k in ('raw', 'rgb', 'depth', 'weights')}
  return scene, o, d, sd, vals, cot, grads, keep


def run_train_static(device, scene, o, d, sd, S, aa, mask_rgb, cot):
  """HIP side: gather -> StaticNetFunction -> CompositeVanillaFunction, loss.backward() through the dyn_train_* kernels."""
  from dynibar_amd import train_static as TS
  sc = to_dev(scene, device)
  fm = scene['static_featmaps'].detach().clone().to(device).requires_grad_(True)  # (a fresh leaf per run: on the emulator's CPU device .to() is the scene's own tensor)
  views = ops.SourceViews(sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], fm.detach())
  od, dd = o.to(device), d.to(device)
  R = o.shape[0]
  pts, z, _ = ops.sample_along_ray(od, dd, sc['depth_range'], S, True)
  from dynibar_amd import train_motion as TM
  rgb_feat, ray_diff, mask, pm = TM.gather(views, fm, R, S, ray_o=od, ray_d=dd, z_vals=z, pix_mask_thresh=1.0)
  prm = {k: v.detach().to(device).requires_grad_(True) for k, v in sd.items()}
  raw = TS.static_raw(prm, (aa, mask_rgb), views, rgb_feat, od, dd, pts, ray_diff, mask)
  out = TS.composite_vanilla(raw, z, pm)
  loss = sum((out[k] * cot[k].to(device)).sum() for k in cot)
  loss.backward()
  grads = {k: v.grad for k, v in prm.items()}
  grads['featmaps'] = fm.grad
  return out, raw, grads


def add_run_spread(sens, g0, more, tag, factor=4.0):
  """sens[k] += factor x (per-element max - min of gradient k over the identical steps g0, *more).  The largest spread goes into the margin table as a
  fraction of 2e-4 of the tensor's largest gradient (informational: the assertion is the caller's assert_close); tensors whose whole gradient is round-off
  (largest entry below 1e-3 of the largest gradient of the step, e.g. rgb_fc.4.bias: the blending softmax is shift invariant) are measured against that floor."""
  out = dict(sens)
  runs = {}
  for k in sens:
    if g0.get(k) is None or any(m.get(k) is None for m in more):
      continue
    runs[k] = torch.stack([cpu(g0[k]).double().reshape(-1)] + [cpu(m[k]).double().reshape(-1) for m in more])
  gmax = max([float(r[0].abs().max()) for r in runs.values()] + [0.0])
  worst, worst_k = 0.0, ''
  for k, r in runs.items():
    sp = (r.max(0).values - r.min(0).values).reshape(sens[k].shape)
    out[k] = sens[k] + factor * sp
    scale = max(float(r[0].abs().max()), 1e-3 * gmax)
    if scale > 0 and float(sp.max()) / scale > worst:
      worst, worst_k = float(sp.max()) / scale, k
  record_margin(f'{tag} run-to-run spread of three identical steps (largest: {worst_k}), fraction of 2e-4 max|g|', worst, 2e-4)
  return out


def check_train_static(device, name='small', S=16, R=None, aa=True, mask_rgb=False, weights='init', seed=0):
  args = (name, S, R, aa, mask_rgb, weights, seed)
  scene, o, d, sd, v_ref, cot, g_ref, keep = train_static_reference(*args)
  assert int(keep.sum()) > 0
  # Conditioning of the reference's own arithmetic, per element: (i) its fp32 values / gradients against the same graph in fp64,
  # (ii) with anti-alias pooling, +-1 ulp jitter of exp() in the weights (e - min_v e), a difference of nearly equal numbers that a
  # different libm already triggers (check_static_net).  Allowance = 3 x (i) + 4 x (ii) on top of the plain limits.
  _, _, _, _, v64, _, g64, _ = train_static_reference(*args, dtype=torch.float64)
  sens_v = {k: 3.0 * (v_ref[k].double() - v64[k]).abs() for k in v_ref}
  sens_g = {k: 3.0 * (g_ref[k].double() - g64[k]).abs() for k in g_ref}
  if aa:
    for js in range(3):
      _, _, _, _, vj, _, gj, _ = train_static_reference(*args, jitter_seed=js)
      for k in v_ref:
        sens_v[k] = torch.maximum(sens_v[k], 4.0 * (vj[k] - v_ref[k]).abs().double())
      for k in g_ref:
        sens_g[k] = torch.maximum(sens_g[k], 4.0 * (gj[k] - g_ref[k]).abs().double())
  out, raw, g = run_train_static(device, scene, o, d, sd, S, aa, mask_rgb, cot)
  tag = f'train {name} aa={int(aa)} mask_rgb={int(mask_rgb)}'
  # (iii) the kernels' own run-to-run spread: weight / bias / feature-map gradients are fp32 atomic sums whose order changes from launch to
  # launch and from box to box, and a gradient that is a sum of cancelling terms moves by more than its last bits.  Two more identical
  # steps; 4 x the per-element spread of the three joins the allowance, so that an unlucky order cannot turn a correct kernel red.
  sens_g = add_run_spread(sens_g, g, [run_train_static(device, scene, o, d, sd, S, aa, mask_rgb, cot)[2] for _ in range(2)], tag)
  assert_close(cpu(raw)[keep][..., :3], v_ref['raw'][keep][..., :3], 1e-4, 0.0, f'{tag} raw rgb', extra=sens_v['raw'][keep][..., :3])
  assert_close(cpu(raw)[keep][..., 3], v_ref['raw'][keep][..., 3], 1e-4, 1e-4, f'{tag} raw sigma', extra=sens_v['raw'][keep][..., 3])
  assert_close(cpu(out['rgb'])[keep], v_ref['rgb'][keep], 1e-4, 0.0, f'{tag} rgb', extra=sens_v['rgb'][keep])
  assert_close(cpu(out['weights'])[keep], v_ref['weights'][keep], 1e-4, 0.0, f'{tag} weights', extra=sens_v['weights'][keep])
  # gradients: fp32-class products, fp32 sums in another order (atomics, split reductions): 2e-4 of the tensor's largest gradient
  # plus 1e-3 relative (+ the conditioning allowance).  A gradient that is a sum of cancelling terms (rgb_fc.4.bias: the blending
  # softmax is shift invariant, so its gradient is analytically zero and both sides hold round-off only) gets an absolute floor of
  # 2e-6 of the largest parameter gradient.
  worst = 0.0
  gmax = max(float(v.abs().max()) for k, v in g_ref.items() if k != 'featmaps')
  for k, ref in g_ref.items():
    assert g[k] is not None, f'{tag}: no gradient for {k}'
    got = cpu(g[k]).reshape(ref.shape)
    scale = float(ref.abs().max())
    assert_close(got, ref, 2e-4 * scale + 2e-6 * gmax, 1e-3, f'{tag} grad {k} (max |g| {scale:.2e})', extra=sens_g[k])
    if scale > 1e-3 * gmax:  # (tensors whose whole gradient is round-off are held to the absolute floor above, not to a ratio)
      worst = max(worst, float((got - ref).abs().max()) / scale)
  return worst


def check_train_recompute(device, name='small', S=16):
  """train_static.RECOMPUTE_HIDDEN (the 256-wide hidden layers recomputed in the backward pass instead of kept): the static step's values
  and gradients are BIT-identical with and without it (the recomputation is the forward launch again), and the dual-branch step -- which
  runs the dynamic net's variant -- still meets its parity limits."""
  from dynibar_amd import train_static as TS
  args = (name, S, None, True, False, 'init', 0)
  scene, o, d, sd, _, cot, _, _ = train_static_reference(*args)
  was = TS.RECOMPUTE_HIDDEN
  try:
    res = []
    for flag in (False, True):
      TS.RECOMPUTE_HIDDEN = flag
      out, raw, g = run_train_static(device, scene, o, d, sd, S, True, False, cot)
      res.append((cpu(raw), {k: cpu(v) for k, v in g.items() if v is not None}))
    assert torch.equal(res[0][0], res[1][0]), 'raw differs with recomputed hidden layers'
    for k, v in res[0][1].items():
      if k in ('featmaps',):  # scattered with atomics: the order of the sums differs from run to run
        assert_close(res[1][1][k], v, 1e-6 * float(v.abs().max()), 1e-5, f'recompute grad {k}')
      else:
        same = torch.equal(res[1][1][k], v)
        # weight gradients are atomic sums too (split reductions): equal to the last bits, not necessarily bit-identical
        assert same or float((res[1][1][k] - v).abs().max()) <= 2e-6 * float(v.abs().max()) + 1e-12, f'recompute grad {k}'
    TS.RECOMPUTE_HIDDEN = True
    check_train_dual(device, name, S=S)
  finally:
    TS.RECOMPUTE_HIDDEN = was


def check_train_composite(device, lengths=(5, 64, 100, 150), R=7, seed=5):
  """The compositing autograd nodes alone (train_static.composite_vanilla / train_dynamic.composite_dual: raw2outputs_vanilla / raw2outputs,
  render_ray.py:134-330) against fp64 autograd through the oracle's functions: ray lengths below, at and above one wavefront of samples
  (the backward kernels walk a ray in chunks of 64 with carried transmittance / prefix sums), every differentiable output weighted."""
  from dynibar_amd import train_static as TS, train_dynamic as TD
  g = torch.Generator().manual_seed(seed)
  for S in lengths:
    raw_dy = torch.cat([torch.rand(R, S, 3, generator=g), torch.randn(R, S, 1, generator=g) * 2.0 - 1.0], dim=2)
    raw_st = torch.cat([torch.rand(R, S, 3, generator=g), torch.randn(R, S, 1, generator=g) * 2.0 - 1.0], dim=2)
    z = torch.sort(torch.rand(R, S, generator=g) * 4.0 + 1.0, dim=1).values
    pm = torch.ones(R, S)
    cots = {k: torch.randn(*shape, generator=g) for k, shape in (('rgb', (R, 3)), ('depth', (R,)), ('weights', (R, S)), ('rgb_static', (R, 3)),
                                                                  ('rgb_dy', (R, 3)), ('weights_dy', (R, S)), ('weights_st', (R, S)))}
    d = lambda t: t.detach().clone().to(device).contiguous()
    # one branch
    x = raw_st.double().requires_grad_(True)
    ref = O.raw2outputs_vanilla(x, z.double(), pm.double())
    sum((ref[k] * cots[k].double()).sum() for k in ('rgb', 'depth', 'weights')).backward()
    xd = d(raw_st).requires_grad_(True)
    out = TS.composite_vanilla(xd, d(z), d(pm))
    sum((out[k] * d(cots[k])).sum() for k in ('rgb', 'depth', 'weights')).backward()
    gmax = float(x.grad.abs().max())
    assert_close(xd.grad, x.grad, 2e-6 * gmax + 1e-7, 2e-5, f'train composite (vanilla) S={S} d raw')
    # two branches
    a, b = raw_dy.double().requires_grad_(True), raw_st.double().requires_grad_(True)
    ref = O.raw2outputs(a, b, z.double(), pm.double(), pm.double())
    keys = ('rgb', 'rgb_static', 'rgb_dy', 'depth', 'weights_dy', 'weights_st', 'weights')
    sum((ref[k] * cots[k].double()).sum() for k in keys).backward()
    ad, bd = d(raw_dy).requires_grad_(True), d(raw_st).requires_grad_(True)
    out = TD.composite_dual(ad, bd, d(z), d(pm), d(pm))
    sum((out[k] * d(cots[k])).sum() for k in keys).backward()
    for name, got, want in (('dy', ad.grad, a.grad), ('st', bd.grad, b.grad)):
      gmax = float(want.abs().max())
      assert_close(got, want, 2e-6 * gmax + 1e-7, 2e-5, f'train composite (two branches) S={S} d raw_{name}')


def check_train_attention(device, lengths=(5, 16, 37, 64, 112, 120), R=3, seed=11):
  """dyn_train_attn / dyn_train_attn_bwd alone (ScaledDotProductAttention of the ray transformer, mlp_network.py:13-31, and its autograd)
  against fp64 torch on ray lengths that are not multiples of four, at the LDS form's limit (112) and beyond it (the global-scratch form);
  some query rows masked (nvalid <= 1: their scores are filled with -1e9, so they attend uniformly and pass no gradient to q / k)."""
  from dynibar_amd import train_static as TS
  from dynibar_amd._lib import call
  g = torch.Generator().manual_seed(seed)
  for S in lengths:
    P = R * S
    qkv = torch.randn(P, 384, generator=g) * 1.5
    nvalid = torch.randint(0, 5, (P,), generator=g).float()
    dout = torch.randn(P, 128, generator=g)
    # reference (fp64): [R, 4, S, 32] heads
    x = qkv.double().requires_grad_(True)
    q, k, v = (x[:, 128 * t:128 * (t + 1)].view(R, S, 4, 32).transpose(1, 2) for t in range(3))
    att = torch.matmul(q / 32 ** 0.5, k.transpose(2, 3))
    att = att.masked_fill((nvalid.view(R, 1, S, 1) > 1) == 0, -1e9)
    prob_ref = F.softmax(att, dim=-1)
    out_ref = torch.matmul(prob_ref, v).transpose(1, 2).reshape(P, 128)
    out_ref.backward(dout.double())
    # product
    d = lambda t: t.detach().clone().to(device).contiguous()
    qkv_d, nv_d, dout_d = d(qkv), d(nvalid), d(dout)
    out = torch.empty(P, 128, device=device); prob = torch.empty(R * 4, S, S, device=device)
    dqkv = torch.empty(P, 384, device=device); dsc = torch.empty(R * 4, S, S, device=device)
    st = TS.stream_of(qkv_d)
    call('dyn_train_attn', TS._p(qkv_d), TS._p(nv_d), R, S, TS._p(out), TS._p(prob), st)
    call('dyn_train_attn_bwd', TS._p(qkv_d), TS._p(nv_d), R, S, TS._p(prob), TS._p(dout_d), TS._p(dsc), TS._p(dqkv), st)
    assert_close(prob.view(R, 4, S, S), prob_ref, 2e-6, 1e-5, f'train attention S={S} probabilities')
    assert_close(out, out_ref, 1e-5, 1e-5, f'train attention S={S} output')
    gref = x.grad
    assert_close(dqkv, gref, 2e-6 * float(gref.abs().max()) + 1e-7, 2e-5, f'train attention S={S} d(q|k|v)')


def check_train_gemm_fuzz(device, n_cases=40, seed=123, max_rows=3000):
  """Random shapes / epilogues through dyn_train_gemm in both kernel forms against fp64: row counts and widths that are not multiples of
  the tile or of four, k tails, padded leading dimensions, bias / per-point addend / row scale / ELU / ReLU (forward), activation derivative
  + column sums (data gradient), split reductions with a scaled reduction index (weight gradient)."""
  import random
  from dynibar_amd import train_static as TS
  from dynibar_amd._lib import call
  rng = random.Random(seed)
  g = torch.Generator().manual_seed(seed)
  try:
    for case in range(n_cases):
      mode = (0, 2, 1)[case % 3]
      call('dyn_train_gemm_mode', mode)
      _GEMM_MODE[0] = mode
      M = rng.choice([1, 7, 127, 128, 129, 300, 1000, rng.randint(2, max_rows)])
      K = rng.choice([1, 3, 4, 31, 32, 33, 64, 70, 104, 128, 136, rng.randint(1, 260)])
      N = rng.choice([1, 4, 35, 36, 64, 128, 129, 132, 200, 256, rng.randint(1, 300)])
      ldx = (K + 3) // 4 * 4 + rng.choice([0, 4])
      ldy = (N + 3) // 4 * 4 + rng.choice([0, 0, 8]) if rng.random() < 0.8 else N
      col0 = rng.choice([0, 0, 4, 5])
      kfull = col0 + K + rng.choice([0, 1, 3])
      V = rng.choice([1, 3, 7])
      X = torch.randn(M, ldx, generator=g).to(device)
      Wf = (torch.randn(N, kfull, generator=g) * 0.3).to(device)
      b = torch.randn(N, generator=g).to(device) if rng.random() < 0.7 else None
      lin = TS._Lin(Wf, b, col0, K)
      st = TS.stream_of(X)
      Wd = Wf[:, col0:col0 + K].double().cpu()
      tag = f'fuzz {case} (mode {mode}, M {M} N {N} K {K} ldx {ldx} ldy {ldy} col0 {col0})'
      # forward
      act = rng.choice([TS.NONE, TS.ELU, 2])
      use_add = rng.random() < 0.4 and ldy % 4 == 0
      use_rs = rng.random() < 0.4
      Pp = torch.randn((M + V - 1) // V, ldy, generator=g).to(device) if use_add else None
      rs = (torch.rand(M, generator=g) * 2.0).to(device) if use_rs else None
      Y = torch.full((M, ldy), float('nan'), device=device)
      TS._gemm(st, TS._p(X), ldx, 1, TS._p(lin.Wop, lin.op_off), lin.op_ld, 1, TS._p(Y), ldy, M, N, K, bias=TS._p(b) if b is not None else None,
               addend=TS._p(Pp) if use_add else None, ld_add=ldy, add_div=V, act=act, rowscale=TS._p(rs) if use_rs else None)
      ref = X[:, :K].double().cpu() @ Wd.T
      if use_rs:
        ref = ref * rs.double().cpu()[:, None]
      if use_add:
        ref = ref + Pp.double().cpu().repeat_interleave(V, 0)[:M, :N]
      if b is not None:
        ref = ref + b.double().cpu()
      ref = torch.nn.functional.elu(ref) if act == TS.ELU else (torch.relu(ref) if act == 2 else ref)
      assert_close(Y[:, :N], ref, 2e-5, 6e-6, tag + ' forward')
      if ldy > N:
        assert bool(torch.isnan(Y[:, N:]).all()), tag + ': forward wrote beyond its columns'
      # backward through _Lin.bwd: weight gradient (+ scaled reduction index where the ring form takes it), data gradient (+ act', column sums)
      dZ = (torch.randn(M, ldy, generator=g) * rng.choice([1.0, 1e-5])).to(device)
      Ys = torch.where(X > 0, X, torch.expm1(X))
      kind = rng.choice([None, TS.ELU, 2])
      dW = torch.zeros_like(Wf)
      dX = torch.full((M, ldx), float('nan'), device=device)
      db = torch.zeros(K, device=device)
      xs = (torch.rand(M, generator=g) * 2.0).to(device) if (mode != 1 and rng.random() < 0.4 and ldx % 4 == 0 and ldy % 4 == 0 and ldy >= (N + 3) // 4 * 4) else None
      try:
        summed = lin.bwd(st, dZ, 0, ldy, Ys, 0, ldx, dW, M, dX, 0, ldx, act_y=(Ys, 0, ldx, kind) if kind else None, dbias=db, x_scale=xs)
      except RuntimeError as e:  # a scaled reduction index where the operands force the tile kernel: refused, not computed wrongly
        assert xs is not None and 'kscale' in str(e), (tag, str(e))
        xs = None
        dW.zero_()
        summed = lin.bwd(st, dZ, 0, ldy, Ys, 0, ldx, dW, M, dX, 0, ldx, act_y=(Ys, 0, ldx, kind) if kind else None, dbias=db)
      big = float(dZ[:, :N].abs().max())
      refx = dZ[:, :N].double().cpu() @ Wd
      if kind:
        y = Ys[:, :K].double().cpu()
        refx = refx * torch.where(y > 0, torch.ones_like(y), (y + 1.0) if kind == TS.ELU else torch.zeros_like(y))
      assert_close(dX[:, :K], refx, 4e-6 * big * max(1.0, N ** 0.5), 6e-6, tag + ' data gradient')
      if summed:
        assert_close(db, refx.sum(0), 4e-6 * big * max(1.0, (N * M) ** 0.5), 1e-5, tag + ' column sums')
      xk = Ys[:, :K].double().cpu() * (xs.double().cpu()[:, None] if xs is not None else 1.0)
      refw = dZ[:, :N].double().cpu().T @ xk
      assert_close(dW[:, col0:col0 + K], refw, 4e-6 * float(refw.abs().max()) + 1e-30, 6e-6, tag + ' weight gradient')
      if col0 > 0:
        assert float(dW[:, :col0].abs().max()) == 0.0, tag + ': weight gradient outside the slice'
  finally:
    call('dyn_train_gemm_mode', 0)
    _GEMM_MODE[0] = 0


def check_train_gemm(device):
  """dyn_train_gemm in its three roles (forward with bias / per-point addend / ELU, data gradient, split weight gradient) vs fp64 matmul,
  on shapes that are not multiples of the tile, at gradient-like magnitudes as well (the bf16 split keeps fp32's exponent range)."""
  import ctypes
  from dynibar_amd import train_static as TS
  g = torch.Generator().manual_seed(5)
  M, N, K, V = 150, 37, 103, 3
  for scale in (1.0, 1e-7):  # gradient-like magnitudes too: the gradient operand is rescaled into the half range inside the kernel
    X = torch.randn(M, 104, generator=g).to(device)
    W = torch.randn(N, K, generator=g).to(device) * 0.3
    b = torch.randn(N, generator=g).to(device)
    Pp = torch.randn(M // V, 40, generator=g).to(device)
    Y = torch.full((M, 40), float('nan'), device=device)
    lin = TS._Lin(W, b)
    if scale == 1.0:
      lin.fwd(TS.stream_of(X), X, 0, 104, Y, 0, 40, M, TS.ELU, addend=Pp, ld_add=40, add_div=V)
      ref = torch.nn.functional.elu(X[:, :K].double().cpu() @ W.double().cpu().T + b.double().cpu() + Pp.double().cpu()[:, :N].repeat_interleave(V, 0))
      assert_close(Y[:, :N], ref, 1e-5, 4e-6, 'train gemm forward')
    dZ = (torch.randn(M, 40, generator=g) * scale * torch.exp(3.0 * torch.randn(M, 1, generator=g))).to(device)  # rows of very different size
    dW = torch.zeros_like(W)
    dX = torch.full((M, 104), float('nan'), device=device)
    lin.bwd(TS.stream_of(X), dZ, 0, 40, X, 0, 104, dW, M, dX, 0, 104)
    big = float(dZ.abs().max())
    assert_close(dX[:, :K], dZ[:, :N].double().cpu() @ W.double().cpu(), 3e-6 * big, 4e-6, f'train gemm data gradient (scale {scale:g})')
    assert_close(dW, dZ[:, :N].double().cpu().T @ X[:, :K].double().cpu(), 1e-5 * big, 4e-6, f'train gemm weight gradient (scale {scale:g})')
  # data gradient with the producing layer's activation derivative folded into the epilogue (act_y): the 4-byte path (103 columns) and
  # the LDS-staged 16-byte path (128 columns), ELU and ReLU, on a row count that is not a multiple of the tile
  for Kin in (103, 128):
    ld = (Kin + 3) // 4 * 4
    Xs = torch.randn(M, ld, generator=g).to(device)          # the saved OUTPUT of the previous layer = this layer's input
    Ws = (torch.randn(N, Kin, generator=g) * 0.3).to(device)
    dZ = torch.randn(M, 40, generator=g).to(device)
    lin = TS._Lin(Ws)
    for kind, name in ((TS.ELU, 'ELU'), (2, 'ReLU')):
      Ys = Xs if kind == 2 else torch.where(Xs > 0, Xs, torch.expm1(Xs))  # a plausible saved output (ELU outputs are > -1)
      dW = torch.zeros_like(Ws)
      dX = torch.full((M, ld), float('nan'), device=device)
      db = torch.zeros(Kin, device=device)
      summed = lin.bwd(TS.stream_of(Xs), dZ, 0, 40, Ys, 0, ld, dW, M, dX, 0, ld, act_y=(Ys, 0, ld, kind), dbias=db)
      y = Ys[:, :Kin].double().cpu()
      der = torch.where(y > 0, torch.ones_like(y), (y + 1.0) if kind == TS.ELU else torch.zeros_like(y))
      big = float(dZ.abs().max())
      ref = (dZ[:, :N].double().cpu() @ Ws.double().cpu()) * der
      assert_close(dX[:, :Kin], ref, 3e-6 * big, 4e-6, f"train gemm data gradient x {name}' ({Kin} columns)")
      # the bias gradient / scale of the result from the GEMM's own tiles: only for 16-byte-aligned result rows (the caller's pass otherwise)
      assert summed == (Kin % 4 == 0), (Kin, summed)
      if summed:
        assert_close(db, ref.sum(0), 3e-5 * big, 4e-6, f"train gemm column sums of the result ({name}, {Kin} columns)")
        key, am = dX._dyn_absmax
        assert key == (0, ld, M, Kin)
        assert abs(float(am) - float(dX[:, :Kin].abs().max())) == 0.0, 'scale of the result'

  # the half-float range of the FORWARD operands (only the gradient operand of the backward GEMMs is rescaled): activations beyond the largest half
  # (65504) are carried by the second part up to 2 x 65504 with the precision of that part alone, and residuals below the smallest normal half
  # (6.1e-5) sit on an absolute floor of 2^-25 per operand.  Pinned here so that the limits are measured facts, not assumptions.
  report = {}
  for tag, lo, hi_, rtol, atol_of_max in (('activations 1e3..6e4', 1e3, 6e4, 4e-6, 2e-6), ('activations 7e4..1.2e5 (second part carries the excess)', 7e4, 1.2e5, 0.0, 4e-4),  # measured: 1.4e-4 (MI355X), 2.3e-4 (emulator: truncating second part)
                                          ('activations 1e-6..6e-5 (subnormal halves)', 1e-6, 6e-5, 0.0, 2e-3)):
    mag = torch.exp(torch.rand(M, 104, generator=g) * (math.log(hi_) - math.log(lo)) + math.log(lo))
    X = (mag * torch.sign(torch.randn(M, 104, generator=g))).to(device)
    W = (torch.randn(N, K, generator=g) * 0.3).to(device)
    Y = torch.full((M, 40), float('nan'), device=device)
    TS._Lin(W).fwd(TS.stream_of(X), X, 0, 104, Y, 0, 40, M, 0)
    ref = X[:, :K].double().cpu() @ W.double().cpu().T
    assert bool(torch.isfinite(Y[:, :N]).all()), f'train gemm forward, {tag}: non-finite result'
    big = float(ref.abs().max())
    assert_close(Y[:, :N], ref, atol_of_max * big, rtol, f'train gemm forward, {tag}')
    report[tag] = float((cpu(Y[:, :N]).double() - ref).abs().max()) / big
  print('  train gemm forward operand range (max error / largest result): ' + '; '.join(f'{k}: {v:.1e}' for k, v in report.items()))

  # split reduction over many rows
  M = 5000
  X = torch.randn(M, 64, generator=g).to(device)
  dZ = torch.randn(M, 1, generator=g).to(device)
  W = torch.randn(1, 64, generator=g).to(device)
  dW = torch.zeros_like(W)
  TS._Lin(W).bwd(TS.stream_of(X), dZ, 0, 1, X, 0, 64, dW, M)
  assert_close(dW, dZ.double().cpu().T @ X.double().cpu(), 2e-4, 2e-6, 'train gemm split weight gradient')
  check_train_gemm_many_tiles(device)
  from dynibar_amd._lib import call
  try:  # the same with every product on the ring form, and with every product on the tile kernel
    for mode, M in ((2, None), (1, 1100)):
      call('dyn_train_gemm_mode', mode)
      _GEMM_MODE[0] = mode
      check_train_gemm_many_tiles(device, M)
  finally:
    call('dyn_train_gemm_mode', 0)
    _GEMM_MODE[0] = 0


_GEMM_MODE = [0]


def mode_has_ring():
  return _GEMM_MODE[0] != 1


def check_train_gemm_many_tiles(device, M=None):
  """The ring form's persistent walk: more tiles than resident workgroups (512 on an MI355X; 7 in the emulator), so that a workgroup passes
  from one tile to the next with the next tile's first operand tiles already requested -- forward with bias, per-point addend and ELU
  over two column tiles and a k range that is not a multiple of the step (72: 3 steps, the last one half empty), data gradient with the
  activation derivative and the column sums, weight gradient over many reduction chunks; weight slices with an odd row stride (the
  padded operand copy)."""
  from dynibar_amd import train_static as TS
  g = torch.Generator().manual_seed(11)
  if M is None:
    M = 70003 if str(device).startswith('cuda') else 1100
  V, K, N, col0, kfull = 7, 70, 200, 140, 211
  P = (M + V - 1) // V
  X = torch.randn(M, 72, generator=g).to(device)
  Wfull = (torch.randn(N, kfull, generator=g) * 0.3).to(device)
  b = torch.randn(N, generator=g).to(device)
  Pp = torch.randn(P, N, generator=g).to(device)
  Y = torch.full((M, N), float('nan'), device=device)
  lin = TS._Lin(Wfull, b, col0, K)
  st = TS.stream_of(X)
  lin.fwd(st, X, 0, 72, Y, 0, N, M, TS.ELU, addend=Pp, ld_add=N, add_div=V)
  Wd = Wfull[:, col0:col0 + K].double().cpu()
  ref = torch.nn.functional.elu(X[:, :K].double().cpu() @ Wd.T + b.double().cpu() + Pp.double().cpu().repeat_interleave(V, 0)[:M])
  assert_close(Y, ref, 1e-5, 4e-6, 'train gemm (many tiles) forward')
  rs = (torch.rand(M, generator=g) * 2.0).to(device)  # a Linear on x * s[row]: the scale applied to the product (vis_fc.0 on x * weight)
  Y.fill_(float('nan'))
  lin.fwd(st, X, 0, 72, Y, 0, N, M, TS.ELU, rowscale=rs)
  ref = torch.nn.functional.elu(rs.double().cpu()[:, None] * (X[:, :K].double().cpu() @ Wd.T) + b.double().cpu())
  assert_close(Y, ref, 1e-5, 4e-6, 'train gemm (many tiles) forward with a row scale')
  dZ = (torch.randn(M, N, generator=g) * 1e-4).to(device)
  dW = torch.zeros_like(Wfull)
  dX = torch.full((M, 72), float('nan'), device=device)
  Ysaved = torch.where(X > 0, X, torch.expm1(X))  # the saved output of the layer that produced X (its first K = 70 columns matter; 72 are read)
  db = torch.zeros(K, device=device)
  lin.bwd(st, dZ, 0, N, Ysaved, 0, 72, dW, M, dX, 0, 72, act_y=(Ysaved, 0, 72, TS.ELU), dbias=db)
  y = Ysaved[:, :K].double().cpu()
  der = torch.where(y > 0, torch.ones_like(y), y + 1.0)
  big = float(dZ.abs().max())
  refx = (dZ.double().cpu() @ Wd) * der
  assert_close(dX[:, :K], refx, 3e-6 * big * 4, 4e-6, 'train gemm (many tiles) data gradient x ELU\'')
  refw = dZ.double().cpu().T @ Ysaved[:, :K].double().cpu()
  assert_close(dW[:, col0:col0 + K], refw, 2e-6 * float(refw.abs().max()), 4e-6, 'train gemm (many tiles) weight gradient')
  assert float(dW[:, :col0].abs().max()) == 0.0 and float(dW[:, col0 + K:].abs().max()) == 0.0, 'weight gradient outside the slice'
  if mode_has_ring():
    # the weight gradient of a Linear that ran on x * s[row]: dW = dZ^T (diag(s) X) from the unscaled X
    dW2 = torch.zeros_like(Wfull)
    lin.bwd(st, dZ, 0, N, Ysaved, 0, 72, dW2, M, x_scale=rs)
    refw2 = dZ.double().cpu().T @ (rs.double().cpu()[:, None] * Ysaved[:, :K].double().cpu())
    assert_close(dW2[:, col0:col0 + K], refw2, 2e-6 * float(refw2.abs().max()), 4e-6, 'train gemm (many tiles) weight gradient with a scaled reduction index')


def oracle_bootstrap_step(kid, w, jitter_seed=None):
  """The static bootstrap iteration on torch-CPU autograd through the oracle -> (loss, {param: grad} + 'featmaps', rgb).
  w [R]: the loss weights (1 - static_mask) * outputs_coarse_ref['mask'] (a forward value of the dual composite)."""
  c = cases.bootstrap_case(kid)
  scene, o, d, uv, _ = cases.scene_case(c['name'])
  S, aa, mr = c['S'], bool(c['aa']), bool(c['mask_rgb'])
  sd = {k: v.clone().requires_grad_(True) for k, v in O.tdict(cases.model_weights(0)['net_coarse_st']).items() if (aa or k != 's')}
  sc = dict(scene)
  fm = scene['static_featmaps'].clone().requires_grad_(True)
  sc['static_featmaps'] = fm
  jit = None
  if jitter_seed is not None:
    Vs = scene['static_src_rgbs'].shape[1]
    jit = (torch.randint(0, 3, (o.shape[0], S, Vs, 1), generator=torch.Generator().manual_seed(50 + jitter_seed)).float() - 1.0) * 6e-8
  out, pts, _ = _oracle_static_graph(sd, sc, o, d, S, aa, mr, jit)
  loss = cases.charbonnier(out['rgb'], c['gt'], w)
  loss.backward()
  grads = {k: v.grad.detach() for k, v in sd.items()}
  grads['featmaps'] = fm.grad.detach()
  return loss.detach(), grads, out['rgb'].detach()


def bootstrap_sensitivity(kid, w, g_ref):
  """4 x how far the oracle's own gradients move under +-1 ulp jitter of exp() in the anti-alias pooling weights (see check_train_static)"""
  sens = {k: torch.zeros_like(v) for k, v in g_ref.items()}
  if cases.bootstrap_case(kid)['aa']:
    for js in range(3):
      gj = oracle_bootstrap_step(kid, w, jitter_seed=js)[1]
      for k in g_ref:
        sens[k] = torch.maximum(sens[k], 4.0 * (gj[k] - g_ref[k]).abs())
  return sens


def check_static_bootstrap_step(device, golden, kid=True):
  """One iteration of the reference's static bootstrap loop (train.py:116-199) exactly as the script drives it: render_rays_mono(...,
  is_train=False) under grad mode on DataParallel-wrapped nn.Modules, Charbonnier loss (criterion.py:58-62, utils.py:32-39) on
  ret['outputs_coarse_st']['rgb'] with the script's static mask, loss.backward(): the gradients that land in the modules' .grad and in
  the static feature maps against the REAL reference's autograd (tests/golden/train_static.npz) and against autograd through the oracle."""
  import types
  from dynibar_amd import projection, render_ray
  c = cases.bootstrap_case(kid)
  name, S, num_vv = c['name'], c['S'], c['num_vv']
  scene, o, d, uv, _ = cases.scene_case(name)
  Vd = scene['src_rgbs'].shape[1]
  fidx, temb, toff = cases.time_args(Vd - num_vv)
  args = types.SimpleNamespace(anti_alias_pooling=c['aa'], mask_rgb=c['mask_rgb'], input_dir=True, input_xyz=False, occ_weights_mode=0)
  model = make_module_model(device, args, shift=5.0)
  batch = make_ray_batch(scene, o, d, uv, device)
  batch['rgb'], batch['static_mask'] = c['gt'].to(device), c['static_mask'].to(device)
  fm_st = scene['static_featmaps'].to(device).requires_grad_(True)
  feat = (scene['featmaps'].to(device), None, fm_st)
  ret = render_ray.render_rays_mono((fidx, None), (temb.to(device), None), (toff, None), batch, model, feat, projection.Projector(device), S, args,
                                    inv_uniform=True, det=True, is_train=False, num_vv=num_vv)
  pred = ret['outputs_coarse_st']['rgb']
  assert pred.requires_grad, "outputs_coarse_st['rgb'] must carry the autograd graph in grad mode"
  w = (1.0 - batch['static_mask']) * ret['outputs_coarse_ref']['mask'].float()
  loss = cases.charbonnier(pred, batch['rgb'], w)
  loss.backward()
  w_ref = torch.from_numpy(golden[f'{name}/w'])
  assert_bitexact(cpu(w), w_ref, f'bootstrap step ({name}) loss weights (static mask x ray mask)')
  loss_o, g_o, _ = oracle_bootstrap_step(kid, w_ref)
  sens = bootstrap_sensitivity(kid, w_ref, g_o)
  params = dict(render_ray._unwrap(model.net_coarse_st).named_parameters())
  got = {k: p.grad for k, p in params.items()}
  got['featmaps'] = fm_st.grad
  for label, loss_ref, g_ref in (('real reference', torch.from_numpy(golden[f'{name}/loss']), {k: torch.from_numpy(golden[f'{name}/grad/{k}']) for k in g_o}),
                                 ('oracle', loss_o, g_o)):
    assert_close(loss, loss_ref, 1e-5, 1e-4, f'bootstrap step ({name}) loss vs {label}')
    gmax = max(float(v.abs().max()) for k, v in g_ref.items() if k != 'featmaps')
    for k, ref in g_ref.items():
      assert got[k] is not None, f'no gradient on {k}'
      scale = float(ref.abs().max())
      assert_close(got[k].reshape(ref.shape), ref, 3e-4 * scale + 3e-6 * gmax, 1e-3, f'bootstrap step ({name}) grad {k} vs {label} (max |g| {scale:.2e})',
                   extra=sens[k].reshape(ref.shape))
  # parameters outside the static branch receive nothing from this loss, like in the reference
  assert all(p.grad is None for p in model.net_coarse_dy.parameters())
  return float(loss)


# ----------------------------------------------------------------------------------------------------------------------
# training, second slice: DynibarDynamic + the two-branch compositing (sample locations fixed: no motion-path gradient yet)
# ----------------------------------------------------------------------------------------------------------------------
def train_dual_reference(name, S, R, weights='init', shift=5.0, seed=0, dtype=torch.float32, jitter_seed=None):
  """Oracle autograd of: gather at the (fixed) motion-displaced points -> DynibarDynamic, gather -> DynibarStatic, raw2outputs +
  raw2outputs_vanilla(raw_dy); gradients w.r.t. both nets' parameters and both feature-map sets."""
  di = dynamic_inputs(name, S, R, weights)
  scene, o, d = di['scene'], di['o'], di['d']
  cv = lambda v: v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v
  sd_dy = {k: v.clone().to(dtype).requires_grad_(True) for k, v in di['W']['net_coarse_dy'].items()}
  sd_st = {k: v.clone().to(dtype).requires_grad_(True) for k, v in di['W']['net_coarse_st'].items()}
  sc = {k: cv(v) for k, v in scene.items()}
  fm_dy, fm_st = scene['featmaps'].clone().to(dtype).requires_grad_(True), scene['static_featmaps'].clone().to(dtype).requires_grad_(True)
  pts, pts_seq, z = cv(di['pts']), cv(di['pts_seq']), cv(di['z'])
  o, d = cv(o), cv(d)
  prev = torch.get_default_dtype()
  torch.set_default_dtype(dtype)
  try:
    rf, rd, mk = O.compute_with_motions(pts, pts_seq, sc['camera'], sc['src_rgbs'], sc['src_cameras'], fm_dy)
    Vd, Vs = rf.shape[2], sc['static_src_rgbs'].shape[1]
    raw_dy = O.dynamic_net(sd_dy, pts, rf, F.normalize(d, dim=-1), rd, torch.zeros(pts.shape[0], S, Vd, 1), mk, cv(di['t_emb']), shift=shift)
    rfs, rds, mks = O.compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], fm_st)
    jit = None
    if jitter_seed is not None:  # +-1 ulp on exp() of the anti-alias pooling weights (see check_static_net)
      jit = ((torch.randint(0, 3, (pts.shape[0], S, Vs, 1), generator=torch.Generator().manual_seed(50 + jitter_seed)).to(dtype) - 1.0) * 6e-8)
    raw_st = O.static_net(sd_st, pts, O.ref_plucker(o, d), O.src_plucker(pts, sc['static_src_cameras']), rfs, F.normalize(d, dim=-1), rds, mks, True, False,
                          exp_jitter=jit)
    pm_dy, pm_st = mk[..., 0].sum(dim=2) > 1, mks[..., 0].sum(dim=2) > 1
    out = O.raw2outputs(raw_dy, raw_st, z, pm_dy, pm_st)
    out_dy = O.raw2outputs_vanilla(raw_dy, z, pm_dy)
  finally:
    torch.set_default_dtype(prev)
  keep = ~(boundary_margin(di['pts_seq'], scene['src_cameras'][0]).any(dim=2).any(dim=1) |
           boundary_margin(di['pts'][None].repeat(Vs, 1, 1, 1), scene['static_src_cameras'][0]).any(dim=2).any(dim=1))
  g = torch.Generator().manual_seed(200 + seed)
  n = o.shape[0]
  k3, k1, ks = keep[:, None].float(), keep.float(), keep[:, None].float()
  cot = {'rgb': torch.randn(n, 3, generator=g) * k3, 'rgb_static': 0.5 * torch.randn(n, 3, generator=g) * k3,
         'rgb_dy': 0.5 * torch.randn(n, 3, generator=g) * k3, 'depth': 0.1 * torch.randn(n, generator=g) * k1,
         'weights_dy': 0.3 * torch.randn(n, S, generator=g) * ks, 'weights_st': 0.3 * torch.randn(n, S, generator=g) * ks,
         'weights': 0.3 * torch.randn(n, S, generator=g) * ks, 'dy_rgb': torch.randn(n, 3, generator=g) * k3}
  loss = sum((out[k] * cot[k].to(dtype)).sum() for k in cot if k != 'dy_rgb') + (out_dy['rgb'] * cot['dy_rgb'].to(dtype)).sum()
  loss.backward()
  grads = {'dy/' + k: v.grad.detach() for k, v in sd_dy.items()}
  grads.update({'st/' + k: v.grad.detach() for k, v in sd_st.items()})
  grads['featmaps_dy'], grads['featmaps_st'] = fm_dy.grad.detach(), fm_st.grad.detach()
  vals = {k: out[k].detach() for k in ('rgb', 'rgb_dy', 'weights', 'weights_dy')}
  vals['raw_dy'], vals['raw_st'] = raw_dy.detach(), raw_st.detach()
  return di, vals, cot, grads, keep


def check_train_dual(device, name='small', S=16, R=None, weights='init', shift=5.0, seed=0):
  """values and every gradient of the two-branch training graph (dynamic + static nets, raw2outputs) on the HIP kernels vs the oracle"""
  from dynibar_amd import train_dynamic as TD, train_static as TS
  di, v_ref, cot, g_ref, keep = train_dual_reference(name, S, R, weights, shift, seed)
  assert int(keep.sum()) > 0
  # conditioning allowance, measured: the same graph through the oracle in fp64 -- 3 x how far the reference's own fp32 autograd is from the exact
  # gradient, per element (the static branch's anti-alias pooling weights are a cancellation; the dynamic branch has none and its allowance is ~0)
  g64 = train_dual_reference(name, S, R, weights, shift, seed, dtype=torch.float64)[3]
  sens = {k: 3.0 * (v.double() - g64[k]).abs() for k, v in g_ref.items()}
  for js in range(3):  # ... and 4 x how far +-1 ulp on exp() of the pooling weights moves the oracle's own gradients (as check_train_static does)
    gj = train_dual_reference(name, S, R, weights, shift, seed, jitter_seed=js)[3]
    for k in g_ref:
      sens[k] = sens[k] + 4.0 * (gj[k] - g_ref[k]).abs().double() / 3.0
  scene = di['scene']
  sc = to_dev(scene, device)
  from dynibar_amd import train_motion as TM

  def hip_step():
    fm_dy = scene['featmaps'].detach().clone().to(device).requires_grad_(True)  # (fresh leaves per run)
    fm_st = scene['static_featmaps'].detach().clone().to(device).requires_grad_(True)
    od, dd, pts, pts_seq, z = (di[k].to(device) for k in ('o', 'd', 'pts', 'pts_seq', 'z'))
    Rn = od.shape[0]
    views_dy = ops.SourceViews(sc['camera'], sc['src_rgbs'], sc['src_cameras'], fm_dy.detach())
    views_st = ops.SourceViews(sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], fm_st.detach())
    rf, _, mk, pm_dy = TM.gather(views_dy, fm_dy, Rn, S, xyz=pts_seq, pts_st=pts, pix_mask_thresh=1.0)
    rfs, rds, mks, pm_st = TM.gather(views_st, fm_st, Rn, S, ray_o=od, ray_d=dd, z_vals=z, pix_mask_thresh=1.0)
    prm_dy = {k: v.detach().to(device).requires_grad_(True) for k, v in di['W']['net_coarse_dy'].items()}
    prm_st = {k: v.detach().to(device).requires_grad_(True) for k, v in di['W']['net_coarse_st'].items()}
    raw_dy = TD.dynamic_raw(prm_dy, shift, rf, dd, pts, mk, di['temb'].to(device))
    raw_st = TS.static_raw(prm_st, (True, False), views_st, rfs, od, dd, pts, rds, mks)
    out = TD.composite_dual(raw_dy, raw_st, z, pm_dy, pm_st)
    out_dy = TS.composite_vanilla(raw_dy, z, pm_dy)
    loss = sum((out[k] * cot[k].to(device)).sum() for k in cot if k != 'dy_rgb') + (out_dy['rgb'] * cot['dy_rgb'].to(device)).sum()
    loss.backward()
    got = {'dy/' + k: v.grad for k, v in prm_dy.items()}
    got.update({'st/' + k: v.grad for k, v in prm_st.items()})
    got['featmaps_dy'], got['featmaps_st'] = fm_dy.grad, fm_st.grad
    return raw_dy, out, got

  raw_dy, out, got = hip_step()
  tag = f'train dual {name}'
  assert_close(cpu(raw_dy)[keep][..., :3], v_ref['raw_dy'][keep][..., :3], 1e-4, 0.0, f'{tag} raw_dy rgb')
  assert_close(cpu(raw_dy)[keep][..., 3], v_ref['raw_dy'][keep][..., 3], 1e-4, 1e-4, f'{tag} raw_dy sigma')
  for k in ('rgb_dy', 'weights_dy'):
    assert_close(cpu(out[k])[keep], v_ref[k][keep], 1e-4, 0.0, f'{tag} {k}')
  # the kernels' own run-to-run spread (fp32 atomic sums in a launch-dependent order) joins the allowance: see check_train_static (iii)
  sens = add_run_spread(sens, got, [hip_step()[2] for _ in range(2)], tag)
  gmax = max(float(v.abs().max()) for k, v in g_ref.items() if not k.startswith('featmaps'))
  worst = 0.0
  for k, ref in g_ref.items():  # every gradient of BOTH branches, one limit
    assert got[k] is not None, f'{tag}: no gradient for {k}'
    scale = float(ref.abs().max())
    # round 5: 8e-5 of the tensor's largest gradient (rounds 2-4: 2e-4).  The old limit had been fitted to a defect: the compositing backward formed
    # its suffix sums as total - prefix, whose coherent rounding along a 200-sample ray put 3e-4 of max|g| into every gradient behind sigma
    # (tools/grad_rootcause.py; `few`, S = 200 used 0.96 of the limit).  With the suffix sums formed directly the same tensors are at 4-6e-6 -- the
    # oracle's own fp32 against fp64 -- and the worst tensor of all six cases uses 0.19 of the OLD limit.
    assert_close(cpu(got[k]).reshape(ref.shape), ref, 8e-5 * scale + 8e-7 * gmax, 4e-4, f'{tag} grad {k} (max |g| {scale:.2e})', extra=sens[k])
    if scale > 1e-3 * gmax:
      worst = max(worst, float((cpu(got[k]).reshape(ref.shape) - ref).abs().max()) / scale)
  return worst



# ----------------------------------------------------------------------------------------------------------------------
# training, third slice: the whole main-loop iteration (train.py:203-467) incl. the motion path
# ----------------------------------------------------------------------------------------------------------------------
MONO_TRAIN_CASE = dict(name='few', S=16, R=4, num_vv=1)


def _digest_close(got, ref_d, what, n, gmax):
  """gradient tensor vs its golden digest (cases.grad_digest): projections within (3e-4 absmax + 3e-6 gmax) sqrt(n) + 2e-3 relative, first
  values likewise (gmax: the largest gradient of any network parameter -- the round-off floor of tensors whose own gradient is tiny)"""
  d = cases.grad_digest(got)
  am = float(ref_d['absmax'][0])
  assert_close(d['proj'], torch.from_numpy(ref_d['proj']), (3e-4 * am + 3e-6 * gmax) * (n ** 0.5) + 1e-12, 2e-3, f'{what} projections (max |g| {am:.2e})')
  assert_close(d['head'], torch.from_numpy(ref_d['head']), 3e-4 * am + 3e-6 * gmax + 1e-12, 2e-3, f'{what} first values')


def oracle_mono_train_step(terms, dtype=torch.float32, case=None, device='cpu'):
  """torch autograd through the oracle's render_rays_mono_train + the restated train.py loss -> (loss, {name: grad}).  dtype float64: the same
  graph in double (the arbiter of how well-conditioned each gradient is); device: where the eager graph runs (the large case runs on the GPU)."""
  c = case or MONO_TRAIN_CASE
  scene, o, d, uv, _ = cases.scene_case(c['name'])
  o, d, uv = o[:c['R']], d[:c['R']], uv[:c['R']]
  sc, fidx, temb, toff = cases.anchor_case(scene, num_vv=c['num_vv'])
  cv = lambda v: (v.to(dtype) if v.is_floating_point() else v).to(device) if isinstance(v, torch.Tensor) else v
  W = {k: {n: cv(v.clone()).requires_grad_(True) for n, v in O.tdict(sd).items()} for k, sd in cases.model_weights_trained().items()}
  W['net_coarse_st'].pop('s', None)
  basis = cv(O.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES).clone()).requires_grad_(True)
  W['trajectory_basis'] = basis
  sc = {k: cv(v) for k, v in sc.items()}
  fms = {k: sc[k].clone().requires_grad_(True) for k in ('featmaps', 'featmaps_anchor', 'static_featmaps')}
  sc.update(fms)
  tgt = {k: cv(v) for k, v in cases.train_batch_targets(c['R']).items()}  # drawn in fp32 whatever the graph's dtype: the same targets for every run
  prev = torch.get_default_dtype()
  torch.set_default_dtype(dtype)
  try:
    ret = O.render_rays_mono_train(W, sc, cv(o), cv(d), cv(uv), fidx, tuple(cv(t) for t in temb), toff, c['S'], True, True, anti_alias_pooling=False, mask_rgb=True,
                                   num_vv=c['num_vv'], dy_shift=5.0)
    loss = cases.mono_train_loss(ret, tgt, terms)
    loss.backward()
  finally:
    torch.set_default_dtype(prev)
  grads = {'basis': basis.grad, 'featmaps_ref': fms['featmaps'].grad, 'featmaps_anchor': fms['featmaps_anchor'].grad,
           'featmaps_static': fms['static_featmaps'].grad}
  for net in ('net_coarse_st', 'net_coarse_dy', 'motion_mlp'):
    for k, v in W[net].items():
      grads[f'{net}.{k}'] = v.grad
  return loss.detach(), grads


def run_mono_train_step(device, terms, case=None):
  """the HIP path: render_ray.render_rays_mono(is_train=True) on DataParallel-wrapped modules under grad mode, the restated loss, backward"""
  import types
  from dynibar_amd import projection, render_ray
  c = case or MONO_TRAIN_CASE
  scene, o, d, uv, _ = cases.scene_case(c['name'])
  o, d, uv = o[:c['R']], d[:c['R']], uv[:c['R']]
  sc, fidx, temb, toff = cases.anchor_case(scene, num_vv=c['num_vv'])
  args = types.SimpleNamespace(anti_alias_pooling=0, mask_rgb=1, input_dir=True, input_xyz=False, occ_weights_mode=0)
  model = make_module_model(device, args, shift=5.0, weights=cases.model_weights_trained())
  batch = make_ray_batch(sc, o, d, uv, device)
  batch['anchor_src_rgbs'], batch['anchor_src_cameras'] = sc['anchor_src_rgbs'].to(device), sc['anchor_src_cameras'].to(device)
  fms = [sc[k].to(device).requires_grad_(True) for k in ('featmaps', 'featmaps_anchor', 'static_featmaps')]
  ret = render_ray.render_rays_mono(fidx, tuple(t.to(device) for t in temb), toff, batch, model, tuple(fms), projection.Projector(device), c['S'], args,
                                    inv_uniform=True, det=True, is_train=True, num_vv=c['num_vv'])
  loss = cases.mono_train_loss(ret, cases.train_batch_targets(c['R']), terms)
  loss.backward()
  grads = {'basis': model.trajectory_basis.grad, 'featmaps_ref': fms[0].grad, 'featmaps_anchor': fms[1].grad, 'featmaps_static': fms[2].grad}
  for net in ('net_coarse_st', 'net_coarse_dy', 'motion_mlp'):
    for k, p in render_ray._unwrap(getattr(model, net)).named_parameters():
      grads[f'{net}.{k}'] = p.grad
  return loss.detach(), grads


def check_train_mono(device, golden, losses=('full', 'flow', 'cycle', 'reg', 'rgb')):
  """One iteration of the reference's main loop (train.py:203-467) through dynibar_amd.render_ray.render_rays_mono: the loss value and the
  gradient of EVERY parameter (DynibarStatic, DynibarDynamic, MotionMLP, trajectory basis) and of the three feature-map sets against the
  REAL reference's autograd (digests in tests/golden/mono_train_grad.npz), for the full loss and for single terms (so that the flow,
  cycle and regularisation routes into MotionMLP / the basis are each visible on their own)."""
  n_checked = 0
  for lname in losses:
    loss, grads = run_mono_train_step(device, cases.MONO_TRAIN_LOSSES[lname])
    assert_close(loss, torch.from_numpy(golden[f'{lname}/loss']), 1e-5, 2e-4, f'mono train [{lname}] loss')
    keys = sorted({k.split('/')[1] for k in golden if k.startswith(lname + '/') and k.count('/') == 2})
    gmax = max(float(golden[f'{lname}/{k}/absmax'][0]) for k in keys if not k.startswith('featmaps'))
    for k in keys:
      ref_d = {dk: golden[f'{lname}/{k}/{dk}'] for dk in ('proj', 'absmax', 'head', 'l1')}
      g = grads.get(k)
      if g is None:
        assert float(ref_d['absmax'][0]) == 0.0, f'mono train [{lname}]: no gradient for {k} but the reference has one (max {float(ref_d["absmax"][0]):.2e})'
        continue
      _digest_close(g, ref_d, f'mono train [{lname}] grad {k}', g.numel(), gmax)
      n_checked += 1
    missing = [k for k, g in grads.items() if g is not None and f'{lname}/{k}/proj' not in golden and float(g.abs().max()) > 0]
    assert not missing, f'mono train [{lname}]: gradients the reference does not produce: {missing[:5]}'
  return n_checked


MONO_TRAIN_LARGE = dict(name='train_large', S=64, R=256, num_vv=3)


def check_train_mono_large(device, case=None):
  """Section 8f-3 at the shape training runs at (configs/train_kid-running.txt: 64 samples, 7 + 3 dynamic views at the reference and at the anchor frame,
  15 static views) with hundreds of rays: the full train.py loss (cases.mono_train_loss, every term) and EVERY gradient -- all parameters of the three nets,
  the trajectory basis, the three feature-map sets -- as FULL tensors against autograd through the oracle run ON THE DEVICE (PyTorch eager fp32), with
  the same graph in fp64 as the arbiter of conditioning: limit = 2e-4 of the tensor's largest gradient + 2e-6 of the largest parameter gradient + 1e-3
  relative + 3 x |oracle fp32 - oracle fp64| per element (how far the reference's own fp32 arithmetic is from the exact gradient).  Weight gradients are
  fp32 atomic sums over hundreds of workgroups, so two identical steps differ in the last bits: the spread of two runs is measured and reported too."""
  c = case or MONO_TRAIN_LARGE
  terms = cases.MONO_TRAIN_LOSSES['full']
  loss_a, g_a = run_mono_train_step(device, terms, case=c)
  loss_b, g_b = run_mono_train_step(device, terms, case=c)
  loss32, g32 = oracle_mono_train_step(terms, torch.float32, case=c, device=device)
  loss64, g64 = oracle_mono_train_step(terms, torch.float64, case=c, device=device)
  assert_close(loss_a, cpu(loss32), 1e-5, 2e-4, f'train large ({c["R"]} rays) loss vs oracle fp32 on the device')
  assert_close(loss_a, cpu(loss64).float(), 1e-5, 2e-4, f'train large ({c["R"]} rays) loss vs oracle fp64')
  keys = [k for k, v in g64.items() if v is not None]
  gmax = max(float(g64[k].abs().max()) for k in keys if not k.startswith('featmaps'))
  worst, spread, n = 0.0, 0.0, 0
  for k in keys:
    ref64 = cpu(g64[k]).double()
    ref32 = cpu(g32[k]).double()
    got, got_b = g_a.get(k), g_b.get(k)
    scale = float(ref64.abs().max())
    if got is None:
      assert scale == 0.0, f'train large: no gradient for {k} but the oracle has one (max {scale:.2e})'
      continue
    got, got_b = cpu(got).double().reshape(ref64.shape), cpu(got_b).double().reshape(ref64.shape)
    cond = 3.0 * (ref32 - ref64).abs()
    assert_close(got, ref64, 2e-4 * scale + 2e-6 * gmax, 1e-3, f'train large grad {k} vs oracle fp64 (max |g| {scale:.2e}, {ref64.numel()} values)', extra=cond)
    assert_close(got, ref32, 2e-4 * scale + 2e-6 * gmax, 1e-3, f'train large grad {k} vs oracle fp32 on the device (max |g| {scale:.2e})', extra=cond)
    if scale > 1e-3 * gmax:
      worst = max(worst, float((got - ref64).abs().max()) / scale)
      sp = float((got - got_b).abs().max()) / scale
      record_margin(f'train large run-to-run spread of {k} (two identical steps; fp32 atomics), fraction of 2e-4 max|g|', sp, 2e-4)
      spread = max(spread, sp)
    n += 1
  assert n >= 90, 'every parameter of the three nets, the basis and the feature maps'
  assert spread < 2e-4, f'two identical training steps differ by {spread:.1e} of the largest gradient'
  print(f'  train large: {n} gradient tensors at {c["R"]} rays x {c["S"]} samples; worst error {worst:.1e} of a tensor\'s largest gradient, run-to-run spread {spread:.1e}')
  return worst, spread
