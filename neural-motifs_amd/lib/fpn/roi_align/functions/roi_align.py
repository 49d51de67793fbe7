"""
RoIAlign with the reference's calling convention (lib/fpn/roi_align/functions/roi_align.py:9-74):
    RoIAlignFunction(aligned_height, aligned_width, spatial_scale)(features, rois)
rois are [n,5] = (image, x1, y1, x2, y2) in image pixels; the normalisation the reference did in Python happens
inside the kernel (bit-identical arithmetic).  Features may be NCHW or channels_last (NHWC memory); the latter is
what the trunk produces and is read coalesced.
"""
from lib.hip_ops import roi_align


class RoIAlignFunction(object):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def __call__(self, features, rois):
        return roi_align(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale)
