"""
Union-box mask rasteriser with the reference's entry point (Cython lib/draw_rectangles/draw_rectangles.pyx:12-22):
    draw_union_boxes(bbox_pairs [N,8], pooling_size, padding=0) -> [N,2,P,P]
numpy in -> numpy out (the reference contract); CUDA tensor in -> CUDA tensor out (no host round trip).  Either way
the masks are drawn by the gfx950 kernel mh_draw_union_boxes.
"""
import numpy as np
import torch

from lib import _hip


def draw_union_boxes(bbox_pairs, pooling_size, padding=0, offset=0.0, channels_last=False):
    assert padding == 0, "Padding>0 not supported yet"
    if isinstance(bbox_pairs, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(bbox_pairs, dtype=np.float32)).cuda()
        return _hip.draw_union_boxes(t, int(pooling_size), offset, channels_last).cpu().numpy()
    return _hip.draw_union_boxes(bbox_pairs.contiguous().float(), int(pooling_size), offset, channels_last)
