"""Generate golden vectors by running the REAL reference (build container only).

  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference/ibrnet read-only (kornia shimmed, tests/refimport.py), loads the seeded weights
of dynibar_amd.synthetic into the reference's own nn.Modules, runs the reference's own functions on the
seeded scenes of tests/cases.py and stores their OUTPUTS in tests/golden/*.npz.  Inputs are not stored: the
tests regenerate them from the same seeds.  Nothing here is imported by the product.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cases  # noqa: E402
import refimport  # noqa: E402

warnings.filterwarnings('ignore')
torch.set_num_threads(8)
ref = refimport.import_reference()
RR, PJ, NET, SR, RI = ref.render_ray, ref.projection, ref.mlp_network, ref.sample_ray, ref.render_image


def npy(x):
  if isinstance(x, torch.Tensor):
    return x.detach().cpu().numpy().copy()
  return np.asarray(x)


def flat(prefix, d, out):
  for k, v in d.items():
    if v is None:
      continue
    if isinstance(v, dict):
      flat(prefix + k + '/', v, out)
    else:
      out[prefix + k] = npy(v)
  return out


def ref_args(anti_alias_pooling=1, mask_rgb=0, occ_weights_mode=0):
  return types.SimpleNamespace(anti_alias_pooling=anti_alias_pooling, mask_rgb=mask_rgb, input_dir=True, input_xyz=False,
                               occ_weights_mode=occ_weights_mode)


def build_ref_model(weights, n_coarse, n_fine, args, shift=0.0):
  def load(mod, sd):
    own = mod.state_dict()  # a DynibarStatic built with anti_alias_pooling=0 has no `s` (mlp_network.py:330-331)
    missing = mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in own}, strict=True)
    return mod.eval()
  m = types.SimpleNamespace()
  m.net_coarse_st = load(NET.DynibarStatic(args, in_feat_ch=32, n_samples=n_coarse), weights['net_coarse_st'])
  m.net_coarse_dy = load(NET.DynibarDynamic(args, in_feat_ch=32, n_samples=n_coarse, shift=shift), weights['net_coarse_dy'])
  m.net_fine_st = load(NET.DynibarStatic(args, in_feat_ch=32, n_samples=n_fine), weights['net_fine_st'])
  m.net_fine_dy = load(NET.DynibarDynamic(args, in_feat_ch=32, n_samples=n_fine), weights['net_fine_dy'])
  m.motion_mlp = load(NET.MotionMLP(num_basis=cases.NUM_BASIS), weights['motion_mlp'])
  m.motion_mlp_fine = load(NET.MotionMLP(num_basis=cases.NUM_BASIS), weights['motion_mlp_fine'])
  m.trajectory_basis = ref.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  m.trajectory_basis_fine = ref.init_dct_basis(cases.NUM_BASIS, cases.NUM_FRAMES)
  return m


def ray_batch_of(scene, o, d, uv):
  return dict(ray_o=o, ray_d=d, uv_grid=uv, camera=scene['camera'], depth_range=scene['depth_range'],
              src_rgbs=scene['src_rgbs'], src_cameras=scene['src_cameras'],
              static_src_rgbs=scene['static_src_rgbs'], static_src_cameras=scene['static_src_cameras'])


def stage_goldens(name, S=64):
  scene, o, d, uv, pix = cases.scene_case(name)
  out = {}
  weights = cases.model_weights(0)
  args = ref_args()
  model = build_ref_model(weights, S, 2 * S, args)
  proj = PJ.Projector('cpu')
  with torch.no_grad():
    for inv in (True, False):
      pts, z, s = RR.sample_along_camera_ray(o, d, scene['depth_range'], S, inv_uniform=inv, det=True)
      out[f'sample/inv{int(inv)}/pts'] = npy(pts); out[f'sample/inv{int(inv)}/z'] = npy(z); out[f'sample/inv{int(inv)}/s'] = npy(s)
    pts, z, s = RR.sample_along_camera_ray(o, d, scene['depth_range'], S, inv_uniform=True, det=True)
    Vs = scene['static_src_rgbs'].shape[1]
    rf, rd, mk = proj.compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), scene['camera'], scene['static_src_rgbs'],
                                           scene['static_src_cameras'], scene['static_featmaps'])
    out['proj_st/rgb_feat'] = npy(rf); out['proj_st/ray_diff'] = npy(rd); out['proj_st/mask'] = npy(mk)
    refc = RR.compute_ref_plucker_coordinate(o, d)
    srcc = RR.compute_src_plucker_coordinate(pts, scene['static_src_cameras'])
    out['plucker/ref'] = npy(refc); out['plucker/src'] = npy(srcc)
    ray_dir = torch.nn.functional.normalize(d, dim=-1)
    for aa, mr in ((1, 0), (0, 1)):
      net = NET.DynibarStatic(ref_args(aa, mr), in_feat_ch=32, n_samples=S)
      net.load_state_dict({k: torch.from_numpy(v) for k, v in weights['net_coarse_st'].items()}, strict=False)
      raw = net.eval()(pts, refc, srcc, rf, ray_dir, rd, mk)
      out[f'static_net/aa{aa}_mr{mr}/raw'] = npy(raw)
    raw_st = model.net_coarse_st(pts, refc, srcc, rf, ray_dir, rd, mk)
    pm_st = mk[..., 0].sum(dim=2) > 1
    flat('vanilla_st/', RR.raw2outputs_vanilla(raw_st, z, pm_st), out)
    # motion + dynamic branch
    fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
    R = pts.shape[0]
    t_emb = temb[None, None, :].repeat(R, S, 1)
    coeff = model.motion_mlp(torch.cat([pts, t_emb], -1).float())
    out['motion/coeff_raw'] = npy(coeff)
    coeff[:, -int(round(S * 0.1)):, :] *= 0.0
    B = cases.NUM_BASIS
    traj = {off: RR.compute_traj_pts(coeff[..., :B], coeff[..., B:2 * B], coeff[..., 2 * B:], model.trajectory_basis[None, None, fidx + off, :])
            for off in [-3, -2, -1, 0, 1, 2, 3]}
    pts_seq = torch.stack([pts + (traj[off] - traj[0]) for off in toff], 0)
    out['motion/pts_seq'] = npy(pts_seq)
    rf_dy, rd_dy, mk_dy = proj.compute_with_motions(pts, pts_seq, scene['camera'], scene['src_rgbs'], scene['src_cameras'], scene['featmaps'])
    out['proj_dy/rgb_feat'] = npy(rf_dy); out['proj_dy/ray_diff'] = npy(rd_dy); out['proj_dy/mask'] = npy(mk_dy)
    tdiff = (torch.from_numpy(np.array(toff)) / float(cases.NUM_FRAMES))[None, None, :, None].expand(R, S, -1, -1)
    raw_dy = model.net_coarse_dy(pts, rf_dy, ray_dir, rd_dy, tdiff, mk_dy, t_emb)
    out['dynamic_net/raw'] = npy(raw_dy)
    pm_dy = mk_dy[..., 0].sum(dim=2) > 1
    comp = RR.raw2outputs(raw_dy, raw_st, z, pm_dy, pm_st)
    flat('composite/', comp, out)
    out['flow/render_flows'] = npy(RR.compute_optical_flow(comp, pts_seq, scene['src_cameras'], uv))
    # inverse-CDF resampling, both parametrisations, det and with injected u
    w = comp['weights'].clone()
    g = torch.Generator().manual_seed(5)
    u = torch.rand(R, S, generator=g)
    for inv in (True, False):
      if inv:
        iz = 1.0 / z
        bins = torch.flip(0.5 * (iz[:, 1:] + iz[:, :-1]), dims=[1]); ww = torch.flip(w[:, 1:-1], dims=[1])
      else:
        bins = 0.5 * (z[:, 1:] + z[:, :-1]); ww = w[:, 1:-1]
      out[f'pdf/inv{int(inv)}/det'] = npy(RR.sample_pdf(bins.clone(), ww.clone(), S, det=True))
      torch.manual_seed(77)
      smp = RR.sample_pdf(bins.clone(), ww.clone(), S, det=False)
      torch.manual_seed(77)
      out[f'pdf/inv{int(inv)}/u'] = npy(torch.rand(R, S))
      out[f'pdf/inv{int(inv)}/rand'] = npy(smp)
    # full eval paths
    rb = ray_batch_of(scene, o, d, uv)
    # chain-level index exactness (protocol b): record what the reference's own sample_pdf sees and returns INSIDE
    # render_rays_mv (its own coarse weights), restate its index computation and prove the restatement on the recorded output
    seen = {}
    real_sample_pdf = RR.sample_pdf

    def recording_sample_pdf(bins, weights, N_samples, det=False):
      seen['bins'], seen['weights'] = bins.clone(), weights.clone()  # sample_pdf mutates `weights` in place (render_ray.py:23)
      seen['out'] = real_sample_pdf(bins, weights, N_samples, det=det)
      return seen['out']

    RR.sample_pdf = recording_sample_pdf
    try:
      ret = RR.render_rays_mv((fidx, None), (temb, None), (toff, None), rb, model, proj,
                              (scene['featmaps'], None, scene['static_featmaps']),
                              (scene['featmaps_fine'], None, scene['static_featmaps_fine']),
                              S, args, inv_uniform=True, N_importance=S, det=True, is_train=False)
    finally:
      RR.sample_pdf = real_sample_pdf
    flat('mv/', {k: v for k, v in ret.items() if isinstance(v, dict)}, out)
    from oracle import ibr_oracle as O
    smp, inds = O.sample_pdf(seen['bins'].clone(), seen['weights'].clone(), S, det=True, return_inds=True)
    assert torch.equal(smp, seen['out']), 'index restatement does not reproduce the reference sample_pdf output bit for bit'
    out['chain/mv_above_inds'] = npy(inds).astype(np.int32)
    ret = RR.render_rays_mono((fidx, None), (temb, None), (toff, None), rb, model,
                              (scene['featmaps'], None, scene['static_featmaps']), proj,
                              S, args, inv_uniform=True, N_importance=0, det=True, is_train=False, num_vv=0)
    flat('mono/', {k: v for k, v in ret.items() if isinstance(v, dict)}, out)
  np.savez_compressed(os.path.join(HERE, f'stages_{name}.npz'), **out)
  print(name, len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def sampler_goldens():
  """RaySamplerSingleImage: all-pixel rays + the RandomState(234) pixel draw (sample_ray.py:8,143-163,237-260)."""
  scene, o, d, uv, pix = cases.scene_case('small')
  H, W = 48, 64
  rs = np.random.RandomState(0)
  data = dict(camera=scene['camera'], rgb_path='x', depth_range=scene['depth_range'], src_rgbs=scene['src_rgbs'],
              src_cameras=scene['src_cameras'], static_src_rgbs=scene['static_src_rgbs'],
              static_src_cameras=scene['static_src_cameras'],
              rgb=torch.from_numpy(rs.rand(1, H, W, 3).astype(np.float32)),
              disp=torch.from_numpy(rs.rand(1, H, W).astype(np.float32)),
              motion_mask=torch.from_numpy((rs.rand(1, H, W) > 0.5).astype(np.float32)),
              static_mask=torch.from_numpy((rs.rand(1, H, W) > 0.5).astype(np.float32)),
              flows=torch.from_numpy(rs.rand(1, 6, H, W, 2).astype(np.float32)),
              masks=torch.from_numpy(rs.rand(1, 6, H, W).astype(np.float32)),
              anchor_camera=scene['camera'])
  SR.rng = np.random.RandomState(234)
  smp = SR.RaySamplerSingleImage(data, 'cpu')
  out = {'rays_o': npy(smp.rays_o), 'rays_d': npy(smp.rays_d), 'uv_grid': npy(smp.uv_grid)}
  for i, mode in enumerate(['uniform', 'center', 'uniform']):
    rb = smp.random_sample(37, mode, 0.8)
    out[f'rand{i}/selected_inds'] = np.asarray(rb['selected_inds'])
    out[f'rand{i}/ray_d'] = npy(rb['ray_d']); out[f'rand{i}/rgb'] = npy(rb['rgb']); out[f'rand{i}/flows'] = npy(rb['flows'])
  smp2 = SR.RaySamplerSingleImage(data, 'cpu', render_stride=2)
  out['stride2/rays_d'] = npy(smp2.rays_d)
  np.savez_compressed(os.path.join(HERE, 'sampler.npz'), **out)
  print('sampler', len(out))


def image_goldens(chunk_size=80, fname='image_nvi.npz'):
  """render_single_image_nvi on a tiny frame, 3 chunks (render_image.py:9-217).  chunk_size = 63: 192 pixels = 3 x 63 + 3, i.e. the frame ends in a
  chunk of exactly 3 rays, whose Pluecker moments the reference crosses over the RAYS (render_ray.py:375, :392: torch.cross without dim)."""
  cfg = dict(seed=4, H=12, W=16, V=7, n_static=8, smooth=True)
  from dynibar_amd import synthetic as syn
  sc = syn.make_scene(**cfg); fine = syn.make_scene(**dict(cfg, tag=1))
  scene = {k: cases.t(v) for k, v in sc.items()}
  data = dict(camera=scene['camera'], rgb_path='x', depth_range=scene['depth_range'], src_rgbs=scene['src_rgbs'],
              src_cameras=scene['src_cameras'], static_src_rgbs=scene['static_src_rgbs'],
              static_src_cameras=scene['static_src_cameras'])
  smp = SR.RaySamplerSingleImage(data, 'cpu')
  rb = smp.get_all()
  weights = cases.model_weights(0)
  args = ref_args()
  model = build_ref_model(weights, 64, 128, args)
  fidx, temb, toff = cases.time_args(7)
  with torch.no_grad():
    ret = RI.render_single_image_nvi((fidx, None), (temb, None), (toff, None), smp, rb, model, PJ.Projector('cpu'), chunk_size, 64, args,
                                     inv_uniform=True, N_importance=64, det=True,
                                     coarse_featmaps=(scene['featmaps'], None, scene['static_featmaps']),
                                     fine_featmaps=(cases.t(fine['featmaps']), None, cases.t(fine['static_featmaps'])), is_train=False)
  out = {}
  for grp in ('outputs_coarse_ref', 'outputs_fine_ref'):
    for k, v in ret[grp].items():
      if isinstance(v, torch.Tensor):
        out[f'{grp}/{k}'] = npy(v)
  np.savez_compressed(os.path.join(HERE, fname), **out)
  print('image', fname, {k: v.shape for k, v in out.items()})


def stress_goldens():
  """BASELINE configs[4]: render_rays_mv with 16 views in both branches and 128 coarse + 128 fine samples (fine pass: 256 per ray)."""
  name, S = 'stress', 128
  scene, o, d, uv, pix = cases.scene_case(name)
  args = ref_args()
  model = build_ref_model(cases.model_weights(0), S, 2 * S, args)
  fidx, temb, toff = cases.time_args(scene['src_rgbs'].shape[1])
  out = {}
  with torch.no_grad():
    ret = RR.render_rays_mv((fidx, None), (temb, None), (toff, None), ray_batch_of(scene, o, d, uv), model, PJ.Projector('cpu'),
                            (scene['featmaps'], None, scene['static_featmaps']),
                            (scene['featmaps_fine'], None, scene['static_featmaps_fine']),
                            S, args, inv_uniform=True, N_importance=S, det=True, is_train=False)
  flat('mv/', {k: v for k, v in ret.items() if isinstance(v, dict)}, out)
  np.savez_compressed(os.path.join(HERE, 'stress_mv.npz'), **out)
  print(name, len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def image_mono_goldens():
  """render_single_image_mono on a tiny frame, 3 chunks, 5 time-offset views + 2 virtual views (render_image.py:220-439)."""
  cfg = dict(seed=4, H=12, W=16, V=7, n_static=8, smooth=True)
  from dynibar_amd import synthetic as syn
  sc = syn.make_scene(**cfg)
  scene = {k: cases.t(v) for k, v in sc.items()}
  data = dict(camera=scene['camera'], rgb_path='x', depth_range=scene['depth_range'], src_rgbs=scene['src_rgbs'],
              src_cameras=scene['src_cameras'], static_src_rgbs=scene['static_src_rgbs'],
              static_src_cameras=scene['static_src_cameras'])
  smp = SR.RaySamplerSingleImage(data, 'cpu')
  rb = smp.get_all()
  args = ref_args()
  model = build_ref_model(cases.model_weights(0), 64, 128, args)
  fidx, temb, toff = cases.time_args(7)
  with torch.no_grad():
    ret = RI.render_single_image_mono((fidx, None), (temb, None), (toff[:5], None), smp, rb, model, PJ.Projector('cpu'), 80, 64, args,
                                      inv_uniform=True, N_importance=0, det=True,
                                      featmaps=(scene['featmaps'], None, scene['static_featmaps']), is_train=False, num_vv=2)
  out = {}
  for grp, d in ret.items():
    if isinstance(d, dict):
      for k, v in d.items():
        if isinstance(v, torch.Tensor):
          out[f'{grp}/{k}'] = npy(v)
  np.savez_compressed(os.path.join(HERE, 'image_mono.npz'), **out)
  print('image_mono', {k: v.shape for k, v in out.items()})


def image_mono_train_goldens():
  """render_single_image_mono(is_train=True) on the tiny frame: adds the anchor group (render_image.py:333-336, 412-437)."""
  cfg = dict(seed=4, H=12, W=16, V=7, n_static=8, smooth=True)
  from dynibar_amd import synthetic as syn
  sc = syn.make_scene(**cfg)
  scene = {k: cases.t(v) for k, v in sc.items()}
  scn, fi, te, to = cases.anchor_case(scene, 2, 1)
  data = dict(camera=scn['camera'], rgb_path='x', depth_range=scn['depth_range'], src_rgbs=scn['src_rgbs'],
              src_cameras=scn['src_cameras'], static_src_rgbs=scn['static_src_rgbs'], static_src_cameras=scn['static_src_cameras'],
              anchor_src_rgbs=scn['anchor_src_rgbs'], anchor_src_cameras=scn['anchor_src_cameras'])
  smp = SR.RaySamplerSingleImage(data, 'cpu')
  rb = smp.get_all()
  args = ref_args()
  model = build_ref_model(cases.model_weights(0), 64, 128, args)
  with torch.no_grad():
    ret = RI.render_single_image_mono(fi, te, to, smp, rb, model, PJ.Projector('cpu'), 80, 64, args, inv_uniform=True, N_importance=0, det=True,
                                      featmaps=(scn['featmaps'], scn['featmaps_anchor'], scn['static_featmaps']), is_train=True, num_vv=2)
  out = {}
  for grp, d in ret.items():
    if isinstance(d, dict):
      for k, v in d.items():
        if isinstance(v, torch.Tensor):
          out[f'{grp}/{k}'] = npy(v)
        elif isinstance(v, list):
          for i, t in enumerate(v):
            out[f'{grp}/{k}#{i}'] = npy(t)
  np.savez_compressed(os.path.join(HERE, 'image_mono_train.npz'), **out)
  print('image_mono_train', len(out), 'arrays')


def mono_train_goldens():
  """render_rays_mono(is_train=True): reference-time pass + cross-time rendering at the anchor (render_ray.py:1099-1270), forward
  values, for an adjacent anchor (occ mode 0 -> full weights) and an anchor two frames away (-> composite-dy weights)."""
  out = {}
  scene, o, d, uv, pix = cases.scene_case('small')
  model = build_ref_model(cases.model_weights(0), 64, 128, ref_args())
  for tag, shift, mode in (('adj', 1, 0), ('far', -2, 0), ('mode1', 1, 1)):
    sc, fi, te, to = cases.anchor_case(scene, 2, shift)
    rb = ray_batch_of(sc, o, d, uv)
    rb['anchor_src_rgbs'] = sc['anchor_src_rgbs']; rb['anchor_src_cameras'] = sc['anchor_src_cameras']
    with torch.no_grad():
      ret = RR.render_rays_mono(fi, te, to, rb, model, (sc['featmaps'], sc['featmaps_anchor'], sc['static_featmaps']), PJ.Projector('cpu'),
                                64, ref_args(occ_weights_mode=mode), inv_uniform=True, N_importance=0, det=True, is_train=True, num_vv=2)
    flat(f'{tag}/', {k: v for k, v in ret.items() if isinstance(v, dict)}, out)
  np.savez_compressed(os.path.join(HERE, 'mono_train.npz'), **out)
  print('mono_train', len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def encoder_goldens():
  """The reference's ResNet feature encoder (feature_network.py:179-311) on seeded image batches, incl. odd sizes (reflect padding, strides)."""
  FN = ref.feature_network
  out = {}
  for name in ('small', 'odd', 'wide'):
    imgs, sd = cases.encoder_case(name)
    net = FN.ResNet(coarse_out_ch=32, fine_out_ch=32, coarse_only=False)
    own = net.state_dict()
    assert all(k in own and tuple(own[k].shape) == tuple(v.shape) for k, v in sd.items())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)  # the decoder layers the forward never runs keep their init
    with torch.no_grad():
      xc, xf = net.eval()(imgs.permute(0, 3, 1, 2))
    out[f'{name}/coarse'] = npy(xc); out[f'{name}/fine'] = npy(xf)
  np.savez_compressed(os.path.join(HERE, 'encoder.npz'), **out)
  print('encoder', {k: v.shape for k, v in out.items()})


def mono_kid_goldens():
  """render_rays_mono with the monocular configs' arguments (configs/train_kid-running.txt:41-42,69: anti_alias_pooling = 0, mask_rgb = 1,
  num_vv = 3; DynibarMono builds the dynamic net with shift = 5.0, model.py:304-309): the real DynibarStatic then has no `s`."""
  out = {}
  scene, o, d, uv, pix = cases.scene_case('kid')
  args = ref_args(anti_alias_pooling=0, mask_rgb=1)
  model = build_ref_model(cases.model_weights(0), 64, 128, args, shift=5.0)
  assert 's' not in model.net_coarse_st.state_dict()
  fidx, temb, toff = cases.time_args(7)
  with torch.no_grad():
    ret = RR.render_rays_mono((fidx, None), (temb, None), (toff, None), ray_batch_of(scene, o, d, uv), model,
                              (scene['featmaps'], None, scene['static_featmaps']), PJ.Projector('cpu'), 64, args, inv_uniform=True,
                              N_importance=0, det=True, is_train=False, num_vv=3)
  flat('mono/', {k: v for k, v in ret.items() if isinstance(v, dict)}, out)
  np.savez_compressed(os.path.join(HERE, 'mono_kid.npz'), **out)
  print('mono_kid', len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def train_static_goldens():
  """One iteration of the reference's static bootstrap loop (train.py:116-199) on the REAL modules with the REAL autograd:
  render_rays_mono(is_train=False) under grad mode, the script's Charbonnier loss on ret['outputs_coarse_st']['rgb'] with its static mask,
  loss.backward(); stores the loss, the mask and the gradient of every DynibarStatic parameter and of the static feature maps, for the
  kid-running argument set (anti_alias_pooling=0, mask_rgb=1, num_vv=3) and the Nvidia one (anti_alias_pooling=1, mask_rgb=0)."""
  out = {}
  for kid in (True, False):
    c = cases.bootstrap_case(kid)
    scene, o, d, uv, pix = cases.scene_case(c['name'])
    args = ref_args(anti_alias_pooling=c['aa'], mask_rgb=c['mask_rgb'])
    model = build_ref_model(cases.model_weights(0), c['S'], 2 * c['S'], args, shift=5.0)
    model.net_coarse_st.train()
    Vd = scene['src_rgbs'].shape[1]
    fidx, temb, toff = cases.time_args(Vd - c['num_vv'])
    fm = scene['static_featmaps'].clone().requires_grad_(True)
    ret = RR.render_rays_mono((fidx, None), (temb, None), (toff, None), ray_batch_of(scene, o, d, uv), model,
                              (scene['featmaps'], None, fm), PJ.Projector('cpu'), c['S'], args, inv_uniform=True,
                              N_importance=0, det=True, is_train=False, num_vv=c['num_vv'])
    w = (1.0 - c['static_mask']) * ret['outputs_coarse_ref']['mask'].float()
    loss = cases.charbonnier(ret['outputs_coarse_st']['rgb'], c['gt'], w)
    loss.backward()
    tag = c['name'] + '/'
    out[tag + 'loss'] = npy(loss)
    out[tag + 'w'] = npy(w)
    out[tag + 'rgb'] = npy(ret['outputs_coarse_st']['rgb'])
    out[tag + 'grad/featmaps'] = npy(fm.grad)
    for k, p in model.net_coarse_st.named_parameters():
      out[tag + 'grad/' + k] = npy(p.grad)
    assert all(p.grad is None for p in model.net_coarse_dy.parameters())
  np.savez_compressed(os.path.join(HERE, 'train_static.npz'), **out)
  print('train_static', len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def mono_train_grad_goldens(name='few', S=16, R=4):
  """The reference's main training iteration (train.py:203-467) on the REAL modules with the REAL autograd: render_rays_mono(is_train=True)
  under grad mode, the script's loss (tests/cases.mono_train_loss: the full sum, and the flow / cycle / regularisation / colour terms on
  their own so that every gradient route is visible), loss.backward().  Stores a digest (cases.grad_digest) of the gradient of every
  parameter of net_coarse_st, net_coarse_dy, motion_mlp, of the trajectory basis and of the three feature-map sets."""
  out = {}
  scene, o, d, uv, pix = cases.scene_case(name)
  o, d, uv = o[:R], d[:R], uv[:R]
  sc, fidx, temb, toff = cases.anchor_case(scene, num_vv=1)
  tgt = cases.train_batch_targets(R)
  args = ref_args(anti_alias_pooling=0, mask_rgb=1)
  for lname, terms in cases.MONO_TRAIN_LOSSES.items():
    model = build_ref_model(cases.model_weights_trained(), S, 2 * S, args, shift=5.0)
    for m in (model.net_coarse_st, model.net_coarse_dy, model.motion_mlp):
      m.train()
    model.trajectory_basis = model.trajectory_basis.clone().requires_grad_(True)
    fms = [sc['featmaps'].clone().requires_grad_(True), sc['featmaps_anchor'].clone().requires_grad_(True), sc['static_featmaps'].clone().requires_grad_(True)]
    batch = ray_batch_of(sc, o, d, uv)
    batch['anchor_src_rgbs'], batch['anchor_src_cameras'] = sc['anchor_src_rgbs'], sc['anchor_src_cameras']
    ret = RR.render_rays_mono(fidx, temb, toff, batch, model, tuple(fms), PJ.Projector('cpu'), S, args, inv_uniform=True, N_importance=0, det=True,
                              is_train=True, num_vv=1)
    loss = cases.mono_train_loss(ret, tgt, terms)
    loss.backward()
    out[f'{lname}/loss'] = npy(loss)
    grads = {'basis': model.trajectory_basis.grad, 'featmaps_ref': fms[0].grad, 'featmaps_anchor': fms[1].grad, 'featmaps_static': fms[2].grad}
    for net in ('net_coarse_st', 'net_coarse_dy', 'motion_mlp'):
      for k, p in getattr(model, net).named_parameters():
        grads[f'{net}.{k}'] = p.grad
    for k, g in grads.items():
      if g is None:
        continue
      for dk, dv in cases.grad_digest(g).items():
        out[f'{lname}/{k}/{dk}'] = npy(dv)
  np.savez_compressed(os.path.join(HERE, 'mono_train_grad.npz'), **out)
  print('mono_train_grad', len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def cross_axis_goldens():
  """The shapes on which render_ray.py:375 / :392 (torch.cross without dim) cross over the views, the rays or the samples instead of xyz: the
  reference's own Pluecker functions, DynibarStatic on their outputs, and the whole static stage of render_rays_mono (vanilla compositing)."""
  out = {}
  weights = cases.model_weights(0)
  proj = PJ.Projector('cpu')
  with torch.no_grad():
    for name, S in cases.CROSS_AXIS_SAMPLES.items():
      scene, o, d, uv, pix = cases.scene_case(name)
      pts, z, s = RR.sample_along_camera_ray(o, d, scene['depth_range'], S, inv_uniform=True, det=True)
      Vs = scene['static_src_rgbs'].shape[1]
      rf, rd, mk = proj.compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), scene['camera'], scene['static_src_rgbs'],
                                             scene['static_src_cameras'], scene['static_featmaps'])
      refc = RR.compute_ref_plucker_coordinate(o, d)
      srcc = RR.compute_src_plucker_coordinate(pts, scene['static_src_cameras'])
      out[f'{name}/plucker/ref'] = npy(refc); out[f'{name}/plucker/src'] = npy(srcc)
      ray_dir = torch.nn.functional.normalize(d, dim=-1)
      for aa, mr in ((1, 0), (0, 1)):
        net = NET.DynibarStatic(ref_args(aa, mr), in_feat_ch=32, n_samples=S)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in weights['net_coarse_st'].items()}, strict=False)
        raw = net.eval()(pts, refc, srcc, rf, ray_dir, rd, mk)
        out[f'{name}/static_net/aa{aa}_mr{mr}/raw'] = npy(raw)
        if aa == 1:
          flat(f'{name}/vanilla_st/', RR.raw2outputs_vanilla(raw, z, mk[..., 0].sum(dim=2) > 1), out)
  np.savez_compressed(os.path.join(HERE, 'cross_axis.npz'), **out)
  print('cross_axis', len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


def camera_format_goldens():
  """Section 8f-4, data side: the reference's OWN pose parsing (llff_data_utils.py) and benchmark dataset class (eval_nvidia.py:26-200) on a synthetic
  scene directory (seeded poses_bounds_cvd.npy, dummy image files).  The packages they import for image decoding / metrics (cv2, imageio, skimage,
  models, configargparse) are not installed here and are stubbed: imread returns a blank image of the scene's size -- only cameras, view ids and depth
  ranges are pinned, never pixels."""
  import shutil
  import sys
  import tempfile
  import types
  refimport.import_reference()
  H, W, N = 288, 512, 36
  blank = np.zeros((H, W, 3), np.uint8)
  for name in ('cv2', 'imageio', 'imageio.v2', 'models', 'skimage', 'skimage.metrics', 'skimage.morphology', 'configargparse'):
    if name not in sys.modules:
      sys.modules[name] = types.ModuleType(name)
      sys.modules[name].__path__ = []  # importable as a package (import skimage.morphology)
  sys.modules['skimage'].morphology = sys.modules['skimage.morphology']
  sys.modules['imageio'].imread = lambda f, **kw: blank
  sys.modules['imageio'].v2 = sys.modules['imageio.v2']
  sys.modules['imageio.v2'].imread = lambda f, **kw: blank
  sys.modules['skimage'].metrics = sys.modules['skimage.metrics']
  import importlib
  llff = importlib.import_module('ibrnet.data_loaders.llff_data_utils')
  ev = importlib.import_module('eval_nvidia')
  rng = np.random.RandomState(7)
  # a plausible forward-facing rig: 12 cameras on a shallow arc taking turns frame by frame (the Nvidia benchmark), LLFF axis convention
  poses = np.zeros((N, 3, 5))
  for i in range(N):
    cam = i % 12
    ang = (cam - 5.5) * 0.04 + rng.randn() * 0.003
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]) @ (np.eye(3) + rng.randn(3, 3) * 0.01)
    poses[i, :, :3] = np.linalg.qr(R)[0]
    poses[i, :, 3] = [(cam - 5.5) * 0.21 + rng.randn() * 0.01, rng.randn() * 0.02, rng.randn() * 0.02]
    poses[i, :, 4] = [1080, 1920, 1480.0 + rng.rand()]
  bds = np.stack([2.0 + rng.rand(N), 30.0 + 10 * rng.rand(N)], 1)
  poses_arr = np.concatenate([poses.reshape(N, 15), bds], 1)
  root = tempfile.mkdtemp(prefix='dynibar_cam_')
  try:
    dense = os.path.join(root, 'Scene', 'dense')
    for sub in ('images', 'images_%dx%d' % (W, H)):
      os.makedirs(os.path.join(dense, sub))
      for i in range(N):
        open(os.path.join(dense, sub, '%05d.png' % i), 'wb').close()
    np.save(os.path.join(dense, 'poses_bounds_cvd.npy'), poses_arr)
    out = {'poses_arr': poses_arr, 'image_hw': np.array([H, W])}
    _, lposes, lbds, _, _, _, scale = llff.load_llff_data(dense, height=H, num_avg_imgs=12, render_idx=10, load_imgs=False)
    out.update({'llff/poses': lposes, 'llff/bds': lbds, 'llff/scale': np.array(scale)})
    K, C = llff.batch_parse_llff_poses(lposes)
    out.update({'llff/intrinsics': K, 'llff/c2w': C})
    vv = np.stack([lposes[:4], lposes[4:8]], 1)  # [T=4, Vv=2, 3, 5]
    out['llff/vv_c2w'] = llff.batch_parse_vv_poses(vv)
    out['llff/vv_in'] = vv
    args = types.SimpleNamespace(folder_path=root, mask_static=False)
    for render_idx in (10, 3, 32):
      ds = ev.DynamicVideoDataset(render_idx, args, ['Scene'])
      for view_idx in (0, 7):
        item = ds[view_idx]
        pre = 'item/%d/%d/' % (render_idx, view_idx)
        for k in ('camera', 'src_cameras', 'static_src_cameras', 'depth_range'):
          out[pre + k] = npy(item[k])
        out[pre + 'nearest_pose_ids'] = np.asarray(item['nearest_pose_ids'])
        out[pre + 'ref_time'] = np.array(item['ref_time'])
        out[pre + 'n_static'] = np.array(item['static_src_rgbs'].shape[0])
  finally:
    shutil.rmtree(root, ignore_errors=True)
  np.savez_compressed(os.path.join(HERE, 'camera_format.npz'), **out)
  print('camera_format', len(out), 'arrays', sum(v.nbytes for v in out.values()) // 1024, 'KiB')


if __name__ == '__main__':
  import sys
  if 'camera_format' in sys.argv[1:]:
    camera_format_goldens()
    sys.exit(0)
  if 'cross_axis' in sys.argv[1:]:
    cross_axis_goldens()
    image_goldens(63, 'image_nvi_tail3.npz')
    sys.exit(0)
  if 'mono_train_grad' in sys.argv[1:]:
    mono_train_grad_goldens()
    sys.exit(0)
  if 'train_static' in sys.argv[1:]:
    train_static_goldens()
    sys.exit(0)
  if 'encoder' in sys.argv[1:]:
    encoder_goldens()
    sys.exit(0)
  if 'mono_kid' in sys.argv[1:]:
    mono_kid_goldens()
    sys.exit(0)
  if 'image_mono_train' in sys.argv[1:]:
    image_mono_train_goldens()
    sys.exit(0)
  if 'image_mono' in sys.argv[1:]:
    image_mono_goldens()
    sys.exit(0)
  if 'mono_train' in sys.argv[1:]:
    mono_train_goldens()
    sys.exit(0)
  if 'stress' in sys.argv[1:]:
    stress_goldens()
    sys.exit(0)
  for n in ('small', 'harsh', 'noise'):
    stage_goldens(n)
  stress_goldens()
  sampler_goldens()
  image_goldens()
  image_mono_goldens()
  image_mono_train_goldens()
  mono_train_goldens()
  mono_kid_goldens()
  encoder_goldens()
  camera_format_goldens()
  cross_axis_goldens()
  image_goldens(63, 'image_nvi_tail3.npz')
