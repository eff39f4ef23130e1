"""-m gpu parity tests: the HIP kernels, called through the C-ABI (libdynibar_hip.so), against the CPU oracle on the same
seeded inputs and against the committed golden fixtures generated from the real reference."""
import os

import numpy as np
import pytest
import torch

import cases
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
  assert torch.cuda.is_available(), 'the -m gpu tests need an MI355X'
  from dynibar_amd import _lib
  _lib.lib()  # fails loudly if the gfx950 library is missing
  return 'cuda:0'


def test_sampling(dev):
  parity.check_sampling(dev, 'small')
  parity.check_sampling(dev, 'harsh', S=128)


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_project_gather(dev, name):
  parity.check_project_gather(dev, name)


def test_composite(dev):
  parity.check_composite(dev, R=37, S=64)
  parity.check_composite(dev, R=1000, S=128, seed=1)
  parity.check_composite(dev, R=5, S=200, seed=2)


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_fine_samples(dev, golden_dir, name):
  g = dict(np.load(os.path.join(golden_dir, f'stages_{name}.npz')))
  parity.check_fine_samples(dev, g)


def test_mlp_engine(dev):
  parity.check_mlp_selftest(dev, rows=1000)


def test_mlp_engine_at_the_edges_of_the_half_float_range(dev):
  """activations of 1e-6..6e-5 (subnormal first parts) and 7e4..1.2e5 (saturated first parts) through the INFERENCE engine's layer loop, as the
  training GEMM's range checks do for dyn_train_gemm; measured errors land in the margin table"""
  parity.check_mlp_selftest_ranges(dev)


@pytest.mark.parametrize('scale', [3e-5, 60.0])
def test_static_net_with_features_at_the_edges_of_the_half_float_range(dev, scale):
  parity.check_static_net_feature_range(dev, scale)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=64), dict(name='harsh', S=40, aa=False, mask_rgb=True),
                                dict(name='noise', S=128), dict(name='small', S=32, R=3), dict(name='small', S=256, R=6), dict(name='harsh', S=160, R=5)])
def test_static_net(dev, kw):
  parity.check_static_net(dev, **kw)


@pytest.mark.parametrize('kw', [dict(name='harsh_many', S=64), dict(name='harsh_many', S=40, aa=False, mask_rgb=True), dict(name='harsh_many', S=64, mask_rgb=True, dark=0.3),
                                dict(name='harsh', S=64, mask_rgb=True, dark=0.5), dict(name='small', S=64, mask_rgb=True, dark=0.3), dict(name='many', S=64, R=3)])
def test_static_net_ragged_rows_and_mask_rgb(dev, kw):
  """The ragged dense-rows flavour (11 and 20 static views: rows with mask 0 are not evaluated) over several workgroups and planning segments, on a scene with
  many out-of-bounds / behind-camera samples (points without any valid view included), and mask_rgb removing valid rows and whole points (black source
  colours) -- also in the lane-segment flavour (8 views), whose blend must use the product mask too."""
  parity.check_static_net(dev, **kw)


@pytest.mark.parametrize('name', list(cases.CROSS_AXIS_SAMPLES))
def test_cross_axis_shapes_follow_the_reference(dev, golden_dir, name):
  """render_ray.py:375 / :392 call torch.cross without dim: with exactly 3 static source views, a chunk of exactly 3 rays or 3 samples per ray the
  reference's Pluecker moments are products over that axis (views before rays before samples), and a DynibarStatic trained on such a shape has learnt
  from them.  The helper exports and the network kernels against the REAL reference's outputs on five such shapes, the network and the whole static
  pass against the oracle (itself pinned to the same outputs)."""
  parity.check_cross_axis(dev, dict(np.load(os.path.join(golden_dir, 'cross_axis.npz'))), name)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=64), dict(name='noise', S=64), dict(name='small', S=64, weights='trained'),
                                dict(name='kid', S=64, aa=False, mask_rgb=True), dict(name='harsh_many', S=64, R=24, weights='trained')])
def test_accuracy_against_float64(dev, kw):
  """A check without a tolerance chosen for the kernels: the static pass against the EXACT (float64) values of the reference's formulas, held to twice
  the fp32 reference's own distance from them (largest error and 99th percentile, per output)."""
  parity.check_accuracy_against_double(dev, **kw)


@pytest.mark.parametrize('name,S', [('edge_1_1_4', 8), ('edge_2_2_4', 8), ('edge_1_2_1', 8), ('edge_4_4_4', 2), ('edge_2_1_2', 65), ('edge_1_1_1', 2)])
def test_extreme_shapes(dev, name, S):
  """One or two source views in either branch, a single ray, two samples per ray, one sample more than a row tile: networks, the whole static pass and
  the dynamic network against the oracle (the reference itself yields NaN at one sample per ray: not a case)."""
  parity.check_static_net(dev, name, S=S)
  parity.check_static_pass(dev, name, S=S)
  parity.check_dynamic_net(dev, name, S=S, shift=5.0)


def test_expected_scene_flow(dev):
  parity.check_expected_scene_flow(dev)
  parity.check_expected_scene_flow(dev, R=4099, S=64, seed=9)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=40), dict(name='kid', S=64, virtual_views=3), dict(name='stress', S=128), dict(name='few', S=33),
                                dict(name='train_large', S=64, R=200)])
def test_trajectory_points_fused_into_gather_and_flows(dev, kw):
  """compute_traj_pts inside the gather kernel and the flows (the eval path: the displaced points [V,R,S,3] never exist) against the materialised form: the gather bit-exact, the flows to 5e-5 px"""
  parity.check_fused_trajectory(dev, **kw)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=32, R=4), dict(name='kid', S=48, R=4), dict(name='train_large', S=32, R=96, weights='trained'),
                                dict(name='stress', S=128, R=3)])
def test_dual_branch_accuracy_against_float64(dev, kw):
  """Both branches and the two-branch compositing against the float64 oracle, held to twice the fp32 oracle's own error (no tolerance of ours)."""
  parity.check_dual_accuracy_against_double(dev, **kw)


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_static_pass(dev, name):
  parity.check_static_pass(dev, name)


def test_static_pass_at_baseline_config0_size_every_ray(dev):
  """BASELINE configs[0] at its stated size -- 512 rays x 64 samples x 8 views, 288 x 512 images, white-noise 32-channel feature maps -- through
  sample -> gather -> DynibarStatic -> composite (render_ray.py:1009-1071's static composition) against the oracle on EVERY ray: the kernels start
  from the reference's own fp32 K.inv(c2w), so no ray is dropped and no projection allowance is added (rgb / weights 1e-4, depth 2e-4 relative,
  ray mask bit-exact)."""
  err = parity.check_static_pass(dev, 'config0', same_matrix=True)
  assert err < 1e-4


def test_static_pass_at_the_bench_shape_every_ray(dev):
  """BASELINE configs[1] (4096 rays x 64 samples x 8 views: the shape the headline metric is quoted on), the bench's own scene and rays, ONE 4096-ray
  pass of the HIP path against the oracle on every one of the 4096 rays (rgb / weights 1e-4, depth 2e-4 relative, ray mask bit-exact)."""
  err = parity.check_static_pass_bench_shape_every_ray(dev)
  assert err < 1e-4


@pytest.mark.parametrize('name', ['small', 'harsh'])
def test_trained_scale_weights(dev, name):
  """The split-product engine at the scale of a trained model (activations of tens, density logits -40..+25, LayerNorm gains ~3):
  network outputs within 1e-4 (+1e-4 relative), rendered colours within 1e-4; the achieved fractions are in the margin report."""
  parity.check_static_net(dev, name, S=64, weights='trained')
  parity.check_dynamic_net(dev, name, S=64, shift=5.0, weights='trained')
  parity.check_motion(dev, name, S=64, weights='trained')
  parity.check_static_pass(dev, name, weights='trained')


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=64, shift=5.0), dict(name='noise', S=128), dict(name='small', S=32, R=3), dict(name='small', S=256, R=6), dict(name='harsh', S=160, R=5, shift=2.0)])
def test_dynamic_net(dev, kw):
  parity.check_dynamic_net(dev, **kw)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=128)])
def test_motion_mlp(dev, kw):
  parity.check_motion(dev, **kw)


def _golden(golden_dir, fn):
  return dict(np.load(os.path.join(golden_dir, fn)))


def test_ray_sampler(dev, golden_dir):
  parity.check_ray_sampler(dev, _golden(golden_dir, 'sampler.npz'))


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_render_rays_mv(dev, golden_dir, name):
  parity.check_render_rays_mv(dev, _golden(golden_dir, f'stages_{name}.npz'), name)


def test_render_rays_mv_stress(dev, golden_dir):
  """BASELINE configs[4]: 16 views in both branches, 128 + 128 samples (the long-ray, two-launch point chain) vs the real reference."""
  parity.check_render_rays_mv(dev, _golden(golden_dir, 'stress_mv.npz'), 'stress', S=128)


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_render_rays_mono(dev, golden_dir, name):
  parity.check_render_rays_mono(dev, _golden(golden_dir, f'stages_{name}.npz'), name)


def test_render_rays_mono_kid_running_config(dev, golden_dir):
  """BASELINE configs[3] arguments on real nn.Modules: anti_alias_pooling=0 / mask_rgb=1 / num_vv=3, DataParallel-wrapped nets without `s`."""
  parity.check_render_rays_mono_kid(dev, _golden(golden_dir, 'mono_kid.npz'))


@pytest.mark.parametrize('tag,shift,mode', [('adj', 1, 0), ('far', -2, 0), ('mode1', 1, 1)])
def test_render_rays_mono_train(dev, golden_dir, tag, shift, mode):
  """render_rays_mono(is_train=True) forward values incl. the anchor-frame cross-time rendering, vs the real reference."""
  parity.check_render_rays_mono_train(dev, _golden(golden_dir, 'mono_train.npz'), tag, shift, mode)


def test_render_single_image_nvi(dev, golden_dir):
  parity.check_render_image_nvi(dev, _golden(golden_dir, 'image_nvi.npz'))


def test_render_single_image_nvi_with_a_three_ray_tail_chunk(dev, golden_dir):
  """192 pixels in chunks of 63: the frame ends in a chunk of exactly 3 rays, whose Pluecker moments the reference crosses over the rays (torch.cross
  without dim, render_ray.py:375 / :392) -- its last three pixels differ from the chunk-80 frame's by 0.01-0.035 in rgb.  The whole frame against the
  REAL reference's frame rendered with that chunk size, every pixel."""
  g80, g63 = _golden(golden_dir, 'image_nvi.npz'), _golden(golden_dir, 'image_nvi_tail3.npz')
  d = np.abs(g80['outputs_coarse_ref/rgb'] - g63['outputs_coarse_ref/rgb']).reshape(-1, 3).max(axis=1)
  assert d[:189].max() < 1e-6 and d[189:].min() > 5e-3, 'the golden pair no longer isolates the 3-ray tail'
  parity.check_render_image_nvi(dev, g63, chunk_size=63)


def test_frame_is_identical_on_one_two_and_three_chunk_streams(dev):
  assert parity.check_chunk_stream_invariance(dev) > 0


def test_full_size_frame_is_identical_on_one_and_two_chunk_streams(dev):
  """288 x 512 rays, chunk 8192: the chunks of the two streams overlap on the device for real (round 6: they did not reproduce until no foreign wave could sit
  beside a one-wave-per-SIMD kernel)."""
  assert parity.check_chunk_stream_invariance_full_size(dev) > 0


def test_render_single_image_mono(dev, golden_dir):
  parity.check_render_image_mono(dev, _golden(golden_dir, 'image_mono.npz'))


def test_render_single_image_mono_train(dev, golden_dir):
  parity.check_render_image_mono_train(dev, _golden(golden_dir, 'image_mono_train.npz'))


def test_full_size_properties(dev):
  """BASELINE configs[1] at full size: chunk invariance (bit-exact), compositing / resampling invariants, oracle spot check."""
  parity.check_full_size_properties(dev)


def test_full_frames_against_the_oracle_on_a_strided_ray_subset(dev):
  """BASELINE configs[2] and configs[3] at the size bench.py times them (288 x 512 rays, chunk 8192): the rendered pixels of every 563rd ray (262 rays)
  against the oracle -- render_single_image_nvi (64 + 64 samples, 7 + 11 views) and render_single_image_mono (kid-running arguments)."""
  worst = parity.check_full_frames_vs_oracle(dev)
  assert len(worst) >= 8


def test_full_size_properties_stress(dev):
  """BASELINE configs[4] at full chunk size (8192 rays x 256 samples x 16 views: the two-launch long-ray point chain, 16-lane view
  segments, ~25 GB of workspace): chunk invariance, compositing / resampling invariants, oracle spot check on 48 rays."""
  parity.check_full_size_properties(dev, R=8192, S=256, V=16, N_importance=64)


@pytest.mark.parametrize('name', ['few', 'many'])
def test_networks_other_segment_widths(dev, name):
  """3 / 4 views (4-lane segments, unpooled base_fc.0) and 13 / 20 views (16- and 32-lane segments)."""
  parity.check_static_net(dev, name, S=64)
  parity.check_dynamic_net(dev, name, S=64)
  parity.check_static_pass(dev, name)


def test_render_rays_mono_virtual_views(dev):
  """num_vv = 2 undisplaced virtual source views appended to the dynamic branch (render_ray.py:988-989), against the oracle."""
  parity.check_render_rays_mono_vv(dev)


@pytest.mark.parametrize('name', ['small', 'odd', 'wide'])
def test_feature_encoder(dev, golden_dir, name):
  """Section 8(f)1: the ResNet feature encoder as HIP implicit-GEMM convolutions vs the real reference's feature maps."""
  parity.check_encoder(dev, _golden(golden_dir, 'encoder.npz'), name)


def test_feature_encoder_feeds_the_gather_in_place(dev):
  parity.check_encoder_feeds_gather(dev)


@pytest.mark.parametrize('name', ['small', 'harsh'])
def test_module_helper_exports(dev, golden_dir, name):
  """sample_pdf, compute_traj_pts, compute_optical_flow, compute_*_plucker_coordinate, fine_render_rays with the reference's signatures."""
  parity.check_module_helpers(dev, _golden(golden_dir, f'stages_{name}.npz'), name)


def test_fp32_class_engine_build(dev):
  """libdynibar_hip_x6.so (6-term bf16 split, fp32-class products): engine self-test at 2e-6, static net parity, and -- the interleaved layer loop
  with six partial products per pair (k_motion_mlp) beside the round-3 loop (k_net_points keeps it in this build) -- MotionMLP and the dynamic
  net, in a subprocess because a process binds one library."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  lib = os.path.join(root, 'dynibar_amd', 'csrc', 'libdynibar_hip_x6.so')
  assert os.path.exists(lib), 'python -m dynibar_amd.build builds both engine variants'
  code = ("import sys; sys.path[:0] = [%r, %r]; import parity; from dynibar_amd import _lib; assert _lib.lib().dyn_mlp_split_terms() == 6; "
          "parity.check_mlp_selftest('cuda:0', 500); e = parity.check_static_net('cuda:0', 'small', S=64); "
          "parity.check_motion('cuda:0', 'small', S=64); parity.check_dynamic_net('cuda:0', 'small', S=64, shift=5.0); print('ok', e)") % (root, os.path.join(root, 'tests'))
  r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DYNIBAR_HIP_LIB=lib), capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


# ---- SURVEY section 8(f)3, first slice: the static bootstrap training step (train.py:116-199) -------------------------------------------
def test_train_gemm(dev):
  parity.check_train_gemm(dev)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=32, R=4), dict(name='kid', S=64, aa=False, mask_rgb=True),
                                dict(name='small', S=48, R=5, weights='trained'), dict(name='many', S=16, R=3), dict(name='cross_views', S=8),
                                dict(name='cross_rays_samples', S=3)])
def test_train_static_step(dev, kw):
  """values of raw / rgb / weights and EVERY gradient (39 or 38 parameters + the static feature maps) of the static bootstrap graph against
  torch autograd through the CPU oracle"""
  parity.check_train_static(dev, **kw)


@pytest.mark.parametrize('kid', [True, False])
def test_static_bootstrap_step_drop_in(dev, golden_dir, kid):
  """train.py:116-199 through render_rays_mono on DataParallel-wrapped modules: loss.backward() fills the modules' .grad; against the
  real reference's autograd gradients (golden) and the oracle's"""
  parity.check_static_bootstrap_step(dev, dict(np.load(os.path.join(golden_dir, 'train_static.npz'))), kid=kid)


@pytest.mark.parametrize('kw', [dict(name='small', S=64), dict(name='harsh', S=32, R=4), dict(name='kid', S=48, R=4), dict(name='many', S=16, R=3),
                                dict(name='stress', S=128, R=3), dict(name='few', S=200, R=2)])
def test_train_dual_branch_step(dev, kw):
  """section 8(f)3, second slice: DynibarDynamic (features gathered at the motion-displaced points) + DynibarStatic + raw2outputs +
  raw2outputs_vanilla: values and the gradients of both nets' parameters and both feature-map sets vs autograd through the oracle"""
  parity.check_train_dual(dev, **kw)


def test_full_training_iteration(dev, golden_dir):
  """section 8(f)3 complete: one iteration of the reference's main loop (train.py:203-467) through render_rays_mono(is_train=True) under grad
  mode with the script's loss -- the gradient of every parameter of DynibarStatic, DynibarDynamic, MotionMLP, of the trajectory basis and of
  the three feature-map sets against the REAL reference's autograd, for the full loss and for its flow / cycle / regularisation / colour terms"""
  n = parity.check_train_mono(dev, dict(np.load(os.path.join(golden_dir, 'mono_train_grad.npz'))))
  assert n > 300


def test_training_loop_reduces_the_loss(dev):
  """both training stages of the reference end to end on this package (tools/train_loop.py: Adam over the three MLPs + the trajectory basis,
  render_rays_mono under grad mode, the script's loss, backward through the HIP kernels): the loss must go down in each stage"""
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  import train_loop
  h = train_loop.run(dev, iters=25, R=256, S=32, log_every=24, quiet=True)
  assert h['bootstrap'][-1] < 0.97 * h['bootstrap'][0], h
  assert h['main'][-1] < 0.97 * h['main'][0], h


@pytest.mark.gpu
def test_training_loop_with_the_feature_encoders(dev):
  """the same loop with feature_net / feature_net_st in it (train.py:264-281): maps recomputed from the source images every iteration by the
  encoder's training form, its parameters in the optimizer -- the loss must go down and the encoders' parameters must move"""
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  import train_loop
  h = train_loop.run(dev, iters=20, R=256, S=32, log_every=19, quiet=True, encoders=True)
  assert h['bootstrap'][-1] < 0.97 * h['bootstrap'][0], h
  assert h['main'][-1] < 0.97 * h['main'][0], h


def test_checkpoint_files_to_rendered_frame(dev, golden_dir, tmp_path):
  """Section 8f-4: torch.save files in the reference's two checkpoint formats -> checkpoint.load_model -> HIP encoders (vs the encoder golden) and
  render_single_image_nvi with the loaded model (vs the real reference's frame)."""
  parity.check_checkpoint_render_chain(dev, _golden(golden_dir, 'image_nvi.npz'), _golden(golden_dir, 'encoder.npz'), tmp_path)


def test_training_iteration_at_the_training_shape(dev):
  """Section 8f-3 at 256 rays x 64 samples x (7 + 3) / (7 + 3) / 15 views: the full train.py loss, every gradient as a full tensor vs autograd through
  the oracle on the device (fp32) with its fp64 twin as the arbiter of conditioning; run-to-run spread of two identical steps."""
  parity.check_train_mono_large(dev)


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_gather_and_passes_from_the_reference_matrices(dev, golden_dir, name):
  """With the reference's own fp32 K.inv(c2w) handed in (Projector(matrix_mode=...) / SourceViews(proj_matrices=...)) nothing separates the gather from
  the reference but a 4-term dot product: no conditioning allowance, no dropped frustum-edge rays -- gather, static pass and render_rays_mv on every ray."""
  flips, worst = parity.check_project_gather_same_matrix(dev, name)
  assert flips == 0
  parity.check_static_pass(dev, name, same_matrix=True)
  parity.check_render_rays_mv(dev, _golden(golden_dir, f'stages_{name}.npz'), name, same_matrix=True)


def test_projector_helper_methods(dev):
  """Projector.inbound / normalize / compute_projections / compute_angle (projection.py:13-101): the reference's helper surface, kernel-backed"""
  parity.check_projector_helpers(dev, 'small')
  parity.check_projector_helpers(dev, 'harsh')


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['small', 'odd'])
def test_feature_encoder_training_form(dev, name):
  """forward with saved activations + backward of the encoder (train.py:272-281 optimises feature_net) vs autograd through the oracle"""
  parity.check_encoder_training(dev, name)


@pytest.mark.gpu
def test_feature_encoder_trains_through_the_gather(dev):
  parity.check_encoder_trains_through_gather(dev)


@pytest.mark.gpu
def test_train_composite_lengths(dev):
  """the compositing autograd nodes alone vs fp64 autograd through the oracle: rays of 5 ... 200 samples (the backward walks a ray in chunks of 64)"""
  parity.check_train_composite(dev, lengths=(5, 64, 100, 150, 200), R=70)


@pytest.mark.gpu
def test_train_attention_lengths(dev):
  """the ray attention's training kernels alone vs fp64 torch: ray lengths 5 ... 112 (LDS form, four lanes per row) and 120 / 200 (global-scratch form)"""
  parity.check_train_attention(dev, lengths=(5, 16, 37, 64, 112, 120, 200), R=5)


@pytest.mark.gpu
def test_train_gemm_random_shapes(dev):
  """both kernel forms of dyn_train_gemm on 90 random shapes / epilogues vs fp64 (+ two large ones: more tiles than resident workgroups)"""
  parity.check_train_gemm_fuzz(dev, n_cases=90)
  parity.check_train_gemm_fuzz(dev, n_cases=3, seed=7, max_rows=90000)


@pytest.mark.gpu
def test_training_with_recomputed_hidden_layers(dev):
  """DYNIBAR_TRAIN_RECOMPUTE=1 / train_static.RECOMPUTE_HIDDEN: memory for time, same numbers."""
  parity.check_train_recompute(dev)
