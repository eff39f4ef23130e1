"""On-disk camera and pose formats on the data side of the path (second half).

The renderer takes 34-float camera vectors ``[h, w, K (4x4 row-major), c2w (4x4 row-major)]`` (``sample_ray.parse_camera``); the
reference builds them from LLFF ``poses_bounds`` arrays.  This module is the host-side counterpart of that construction, so that a scene
directory of the Nvidia benchmark can be turned into ray batches without the reference's data loaders:

  * ``parse_llff_pose`` / ``batch_parse_llff_poses`` / ``batch_parse_vv_poses`` -- ibrnet/data_loaders/llff_data_utils.py:14-55
    (LLFF [3,5] pose -> 4x4 intrinsics + OpenCV-convention 4x4 camera-to-world);
  * ``load_poses_bounds``   -- llff_data_utils.py:58-62, 107-109, 246-270 (array layout, image size written into the hwf column, axis order
    fix, bound-based rescale, recentring on the average pose);
  * ``pack_camera`` / ``unpack_camera`` -- eval_nvidia.py:81-83, 136-140, 176-178 (the 34-vector);
  * ``nvidia_eval_view_ids`` -- eval_nvidia.py:92-121 (the 7 temporal neighbours and the per-viewpoint static views of a benchmark frame);
  * ``nvidia_eval_cameras`` -- what ``DynamicVideoDataset.__getitem__`` (eval_nvidia.py:72-200) returns apart from the pixels.

Pure numpy, double precision where the reference is (its arrays are float64 until the final ``astype(np.float32)``); pinned by
tests/golden/camera_format.npz, which tests/golden/make_golden.py writes from the reference's own functions and dataset class.
"""
from __future__ import annotations

import collections

import numpy as np


def parse_llff_pose(pose):
  """LLFF [3,5] pose ([R | t | hwf]) -> (intrinsics [4,4], c2w [4,4]); y and z axes flipped to the OpenCV convention (llff_data_utils.py:14-26)."""
  pose = np.asarray(pose)
  h, w, f = pose[:3, -1]
  c2w = np.eye(4)
  c2w[:3] = pose[:3, :4]
  c2w[:, 1:3] *= -1
  intrinsics = np.array([[f, 0, w / 2.0, 0], [0, f, h / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
  return intrinsics, c2w


def batch_parse_llff_poses(poses):
  """[N,3,5] -> (intrinsics [N,4,4], c2w [N,4,4])  (llff_data_utils.py:29-40)."""
  parsed = [parse_llff_pose(p) for p in poses]
  return np.stack([k for k, _ in parsed]), np.stack([c for _, c in parsed])


def batch_parse_vv_poses(poses):
  """Virtual-view poses of the monocular loader, [T,Vv,3,5] -> c2w [T,Vv,4,4]  (llff_data_utils.py:43-55)."""
  return np.stack([np.stack([parse_llff_pose(p)[1] for p in pose]) for pose in poses])


def _normalize(x):
  return x / np.linalg.norm(x)


def _viewmatrix(z, up, pos):
  vec2 = _normalize(z)
  vec0 = _normalize(np.cross(up, vec2))
  vec1 = _normalize(np.cross(vec2, vec0))
  return np.stack([vec0, vec1, vec2, pos], 1)


def poses_avg(poses):
  """llff_data_utils.py:142-151."""
  hwf = poses[0, :3, -1:]
  center = poses[:, :3, 3].mean(0)
  vec2 = _normalize(poses[:, :3, 2].sum(0))
  up = poses[:, :3, 1].sum(0)
  return np.concatenate([_viewmatrix(vec2, up, center), hwf], 1)


def recenter_poses(poses):
  """llff_data_utils.py:171-183: express every pose in the frame of the average pose."""
  out = poses + 0
  bottom = np.reshape([0, 0, 0, 1.0], [1, 4])
  c2w = np.concatenate([poses_avg(poses)[:3, :4], bottom], -2)
  full = np.concatenate([poses[:, :3, :4], np.tile(bottom[None], [poses.shape[0], 1, 1])], -2)
  full = np.linalg.inv(c2w) @ full
  out[:, :3, :4] = full[:, :3, :4]
  return out


def load_poses_bounds(poses_arr, image_hw, bd_factor=0.75, recenter=True):
  """``poses_bounds*.npy`` ([N,17]) + the size of the images on disk -> (poses [N,3,5] float32, bds [N,2] float32, scale), as
  ``load_llff_data`` prepares them (llff_data_utils.py:58-62 layout, :107-109 image size into the hwf column, :246-263 axis order,
  rescale by 1 / (min bound * bd_factor), recentre)."""
  poses_arr = np.array(poses_arr, copy=True)  # the image size is written into the hwf column below: never through a view of the caller's array
  poses = poses_arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
  bds = poses_arr[:, -2:].transpose([1, 0])
  poses[:2, 4, :] = np.array(image_hw[:2]).reshape([2, 1])
  poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
  poses = np.moveaxis(poses, -1, 0).astype(np.float32)
  bds = np.moveaxis(bds, -1, 0).astype(np.float32)
  scale = 1.0 if bd_factor is None else 1.0 / (bds.min() * bd_factor)
  poses[:, :3, 3] *= scale
  bds *= scale
  if recenter:
    poses = recenter_poses(poses)
  return poses.astype(np.float32), bds, scale


def pack_camera(h, w, intrinsics, c2w):
  """The 34-float camera vector [h, w, K.flatten(), c2w.flatten()] (eval_nvidia.py:81-83)."""
  return np.concatenate(([h, w], np.asarray(intrinsics).flatten(), np.asarray(c2w).flatten())).astype(np.float32)


def unpack_camera(camera):
  """34-vector -> (h, w, K [4,4], c2w [4,4])  (the inverse of pack_camera; sample_ray.py:11-16 does the same on tensors)."""
  camera = np.asarray(camera)
  return int(camera[0]), int(camera[1]), camera[2:18].reshape(4, 4), camera[18:34].reshape(4, 4)


def nvidia_eval_view_ids(render_idx, num_frames, num_imgs_per_cycle=12):
  """(nearest_pose_ids [7], static_pose_ids) of benchmark frame ``render_idx`` (eval_nvidia.py:92-121): the temporal neighbours
  render_idx-3 .. render_idx+3, and -- the benchmark's 12 cameras take turns frame by frame -- for every OTHER camera the frame of it
  nearest in time."""
  nearest = np.sort([render_idx + offset for offset in [1, 2, 3, 0, -1, -2, -3]])
  by_cam = collections.defaultdict(list)
  for i in range(num_frames):
    if i % num_imgs_per_cycle == render_idx % num_imgs_per_cycle:
      continue
    by_cam[i % num_imgs_per_cycle].append(i)
  static = [ids[int(np.argmin(np.abs(np.array(ids) - render_idx)))] for ids in by_cam.values()]
  return nearest, np.sort(static)


def nvidia_eval_cameras(poses, bds, render_idx, view_idx, image_hw=None):
  """Everything ``DynamicVideoDataset(render_idx, ...)[view_idx]`` returns except the pixels (eval_nvidia.py:72-200): the target camera
  (pose of frame ``view_idx``), the 7 dynamic and the static source cameras, the depth range and the reference time.
  poses, bds: as ``load_poses_bounds`` returns them; image_hw: size of the source images (default: the poses' own h, w)."""
  intrinsics, c2w = batch_parse_llff_poses(poses)
  h, w = poses[0][:2, -1]
  src_hw = (int(h), int(w)) if image_hw is None else tuple(image_hw[:2])
  near_depth, far_depth = np.min(bds), np.max(bds) + 15.0
  nearest, static = nvidia_eval_view_ids(render_idx, poses.shape[0])
  cam_of = lambda i: pack_camera(src_hw[0], src_hw[1], intrinsics[i], c2w[i])
  return {
      'camera': pack_camera(int(h), int(w), intrinsics[view_idx], c2w[view_idx]),
      'src_cameras': np.stack([cam_of(i) for i in nearest]),
      'static_src_cameras': np.stack([cam_of(i) for i in static]),
      'depth_range': np.array([near_depth * 0.9, far_depth * 1.5], np.float32),
      'ref_time': float(render_idx / float(poses.shape[0])),
      'id': render_idx,
      'nearest_pose_ids': nearest,
      'static_pose_ids': static,
  }
