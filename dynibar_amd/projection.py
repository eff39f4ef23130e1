"""Drop-in for the reference's ``ibrnet/projection.py``: ``Projector(device).compute_with_motions(...)`` with the reference's
argument order and return shapes (reference projection.py:10, :103-176), backed by the fused ``k_project_gather`` kernel.

Per source-view set the projector prepares, once, the K.inv(c2w) matrices and channels-last feature maps (``ops.SourceViews``)
and reuses them for every ray chunk of the target view: the cache is keyed on the identity and version of the tensors the
caller passes (the reference passes the same ``featmaps`` / ``src_cameras`` objects for all chunks, render_image.py:68-117).
"""
from __future__ import annotations

import os

import torch

from . import ops

DEFAULT_MATRIX_MODE = 'torch'  # see Projector.__init__ (env DYNIBAR_MATRIX_MODE overrides)


class Projector(object):

  def __init__(self, device, matrix_mode=None):
    """matrix_mode -- how the per-view projection matrix K . inv(c2w) (projection.py:42-47) is formed, once per source-view set:
    'torch' (default): by ``torch.inverse`` + ``bmm`` in fp32 on the cameras' device -- the very library call the reference makes, so the matrices
    are bit-for-bit what a reference run on this device computes, and from the same matrices the gather kernel's outputs are bit-for-bit the
    reference's (tests/parity.check_project_gather_same_matrix);
    'exact': in double inside k_prepare_cameras, rounded once to fp32 (what ops.SourceViews does when no matrices are handed in);
    a callable(train_cameras [V,34]) -> [V,4,4]: the caller's own matrices (the parity tests hand in the CPU reference's).
    The fp32 inverse is not canonical -- LAPACK on a CPU, MAGMA / rocSOLVER on a GPU and the exact inverse differ in the last bits -- which is
    the only reason pixel locations of two reference runs on different devices differ; everything downstream follows the reference's arithmetic."""
    if matrix_mode is None:
      matrix_mode = os.environ.get('DYNIBAR_MATRIX_MODE', DEFAULT_MATRIX_MODE)
    assert matrix_mode in ('exact', 'torch') or callable(matrix_mode)  # a callable(train_cameras [V,34]) -> [V,4,4] supplies the matrices (tests)
    self.device = device
    self.matrix_mode = matrix_mode
    self._views = {}

  # ---- the reference's helper methods (projection.py:13-101); the render path does not call them (the gather kernel fuses all four) ----
  def _matrices(self, train_cameras):
    cams = train_cameras.reshape(-1, 34).float()
    if self.matrix_mode == 'torch':
      return cams[:, 2:18].reshape(-1, 4, 4).bmm(torch.inverse(cams[:, -16:].reshape(-1, 4, 4)))
    return self.matrix_mode(cams) if callable(self.matrix_mode) else None

  def inbound(self, pixel_locations, h, w):
    """projection.py:13-20 (a comparison of the caller's tensor: plain tensor expressions, no kernel needed)"""
    return (pixel_locations[..., 0] <= w - 1.0) & (pixel_locations[..., 0] >= 0) & (pixel_locations[..., 1] <= h - 1.0) & (pixel_locations[..., 1] >= 0)

  def normalize(self, pixel_locations, h, w):
    """projection.py:22-30"""
    resize_factor = torch.tensor([w - 1.0, h - 1.0]).to(pixel_locations.device)[None, None, :]
    return 2 * pixel_locations / resize_factor - 1.0

  def compute_projections(self, xyz, train_cameras):
    """projection.py:32-59: xyz [V,...,3], train_cameras [V,34] -> (pixel_locations [V,...,2], in-front mask [V,...] bool); k_project_points"""
    pix, front, _ = ops.project_points(train_cameras, xyz, proj_matrices=self._matrices(train_cameras))
    return pix, front

  def compute_angle(self, xyz_st, xyz, query_camera, train_cameras):
    """projection.py:61-101: -> [V,...,4] = [unit(a - b), a . b]; k_project_points"""
    return ops.project_points(train_cameras, xyz, xyz_st=xyz_st, query_camera=query_camera, want_pix=False)[2]

  def source_views(self, query_camera, train_imgs, train_cameras, featmaps):
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (query_camera, train_imgs, train_cameras, featmaps))
    v = self._views.get(key)
    if v is None:
      # Training hands over NEW featmaps tensors every iteration (the encoder's outputs under grad mode; an iteration uses three or four
      # view sets): entries of earlier iterations can never be hit again, so under grad mode the cache is bounded to about one iteration's
      # sets instead of keeping up to 17 (maps, repacked copy) pairs alive.
      if len(self._views) > (5 if featmaps.requires_grad else 16):
        if featmaps.is_cuda:
          torch.cuda.synchronize(featmaps.device)  # chunks in flight on other streams (render_image.CHUNK_STREAMS) may still read the entries being dropped
        self._views.clear()
      P = self._matrices(train_cameras[0])
      v = ops.SourceViews(query_camera, train_imgs, train_cameras, featmaps, proj_matrices=P)
      # keep the keyed tensors alive so that a recycled address cannot alias a stale entry
      v._key_refs = (query_camera, train_imgs, train_cameras, featmaps)
      # an entry is prepared on the stream of the chunk that first asks for it and may be read by chunks on other streams: readers wait for this event
      # (normally chunk 0 fills the cache before the chunk streams fork; this also orders a miss in a later chunk)
      v._ready = None
      if featmaps.is_cuda:
        v._ready = torch.cuda.Event()
        v._ready.record(torch.cuda.current_stream(featmaps.device))
      self._views[key] = v
    elif v._ready is not None:
      torch.cuda.current_stream(featmaps.device).wait_event(v._ready)
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
