"""Drop-in for the reference's ``ibrnet/projection.py``: ``Projector(device).compute_with_motions(...)`` with the reference's
argument order and return shapes (reference projection.py:10, :103-176), backed by the fused ``k_project_gather`` kernel.

Per source-view set the projector prepares, once, the K.inv(c2w) matrices and channels-last feature maps (``ops.SourceViews``)
and reuses them for every ray chunk of the target view: the cache is keyed on the identity and version of the tensors the
caller passes (the reference passes the same ``featmaps`` / ``src_cameras`` objects for all chunks, render_image.py:68-117).
"""
from __future__ import annotations

import torch

from . import ops


class Projector(object):

  def __init__(self, device):
    self.device = device
    self._views = {}

  def source_views(self, query_camera, train_imgs, train_cameras, featmaps):
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (query_camera, train_imgs, train_cameras, featmaps))
    v = self._views.get(key)
    if v is None:
      # Training hands over NEW featmaps tensors every iteration (the encoder's outputs under grad mode; an iteration uses three or four
      # view sets): entries of earlier iterations can never be hit again, so under grad mode the cache is bounded to about one iteration's
      # sets instead of keeping up to 17 (maps, repacked copy) pairs alive.
      if len(self._views) > (5 if featmaps.requires_grad else 16):
        self._views.clear()
      v = ops.SourceViews(query_camera, train_imgs, train_cameras, featmaps)
      # keep the keyed tensors alive so that a recycled address cannot alias a stale entry
      v._key_refs = (query_camera, train_imgs, train_cameras, featmaps)
      self._views[key] = v
    return v

  def compute_with_motions(self, xyz_st, xyz, query_camera, train_imgs, train_cameras, featmaps):
    """xyz_st [R,S,3], xyz [V,R,S,3], query_camera [1,34], train_imgs [1,V,H,W,3], train_cameras [1,V,34], featmaps [V,F,Hf,Wf]
    -> rgb_feat_sampled [R,S,V,3+F], ray_diff [R,S,V,4], mask [R,S,V,1]."""
    assert (train_imgs.shape[0] == 1) and (train_cameras.shape[0] == 1) and (query_camera.shape[0] == 1), \
        'only support batch_size=1 for now'
    views = self.source_views(query_camera, train_imgs, train_cameras, featmaps)
    R, S = xyz_st.shape[:2]
    assert xyz.shape[0] == views.V and tuple(xyz.shape[1:3]) == (R, S)
    return ops.project_gather(views, R, S, pts_st=xyz_st, xyz=xyz)
