"""Section 8f-4, data side: dynibar_amd.camera_format (LLFF pose parsing, the 34-float camera vector, the Nvidia benchmark's view selection) against
tests/golden/camera_format.npz, which make_golden.py wrote from the reference's own llff_data_utils functions and DynamicVideoDataset class."""
import os

import numpy as np
import pytest

from dynibar_amd import camera_format as CF

G = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'camera_format.npz')))


def test_load_poses_bounds_matches_load_llff_data():
  poses, bds, scale = CF.load_poses_bounds(G['poses_arr'].copy(), G['image_hw'])
  assert poses.dtype == np.float32 and bds.dtype == np.float32
  assert np.array_equal(poses, G['llff/poses']) and np.array_equal(bds, G['llff/bds']) and scale == float(G['llff/scale'])


def test_pose_parsing():
  K, C = CF.batch_parse_llff_poses(G['llff/poses'])
  assert np.array_equal(K, G['llff/intrinsics']) and np.array_equal(C, G['llff/c2w'])
  k0, c0 = CF.parse_llff_pose(G['llff/poses'][3])
  assert np.array_equal(k0, K[3]) and np.array_equal(c0, C[3])
  assert np.array_equal(CF.batch_parse_vv_poses(G['llff/vv_in']), G['llff/vv_c2w'])


@pytest.mark.parametrize('render_idx', [10, 3, 32])
@pytest.mark.parametrize('view_idx', [0, 7])
def test_benchmark_item_cameras(render_idx, view_idx):
  poses, bds, _ = CF.load_poses_bounds(G['poses_arr'].copy(), G['image_hw'])
  item = CF.nvidia_eval_cameras(poses, bds, render_idx, view_idx, image_hw=G['image_hw'])
  pre = 'item/%d/%d/' % (render_idx, view_idx)
  for k in ('camera', 'src_cameras', 'static_src_cameras', 'depth_range'):
    assert item[k].dtype == np.float32 and np.array_equal(item[k], G[pre + k]), k
  assert np.array_equal(item['nearest_pose_ids'], G[pre + 'nearest_pose_ids']) and item['ref_time'] == float(G[pre + 'ref_time'])
  assert len(item['static_pose_ids']) == int(G[pre + 'n_static']) == 11, 'the benchmark has 12 cameras: 11 static source views'


def test_camera_vector_round_trip_and_parse_camera():
  import torch
  from dynibar_amd.sample_ray import parse_camera
  cam = G['item/10/0/camera']
  h, w, K, c2w = CF.unpack_camera(cam)
  assert np.array_equal(CF.pack_camera(h, w, K, c2w), cam)
  W_, H_, Kt, Ct = parse_camera(torch.from_numpy(cam)[None])  # the renderer's own reader (sample_ray.py:11-16)
  assert (int(H_), int(W_)) == (h, w) and np.array_equal(Kt[0].numpy(), K) and np.array_equal(Ct[0].numpy(), c2w)
