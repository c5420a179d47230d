"""Typed views into the opaque forward state, for parity tests only (b3gs_debug_views)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _slice(buf: torch.Tensor, ptr: int, count: int, dtype: torch.dtype) -> torch.Tensor:
    off = ptr - buf.data_ptr()
    nbytes = count * torch.empty(0, dtype=dtype).element_size()
    assert 0 <= off and off + nbytes <= buf.numel(), "view outside the buffer"
    return buf[off:off + nbytes].view(dtype)


def state_views(P: int, W: int, H: int, num_rendered: int, geom: torch.Tensor, binning: torch.Tensor,
                img: torch.Tensor) -> dict:
    L = _lib.lib()
    v = _lib.B3gsDebugViews()
    has_bin = binning is not None and binning.numel() > 1 and num_rendered > 0
    rc = L.b3gs_debug_views(P, W, H, num_rendered, geom.data_ptr(), binning.data_ptr() if has_bin else None,
                            img.data_ptr(), C.byref(v))
    _lib.check(rc, "b3gs_debug_views")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(tiles_touched=_slice(geom, v.tiles_touched, P, torch.int32),
               depth_bits=_slice(geom, v.depths, P, torch.int32),
               records=_slice(geom, v.records, P * 16, torch.float32).view(P, 16),
               ranges=_slice(img, v.ranges, tiles * 2, torch.int32).view(tiles, 2),
               final_T=_slice(img, v.final_T, W * H, torch.float32).view(H, W),
               n_contrib=_slice(img, v.n_contrib, W * H, torch.int32).view(H, W))
    if has_bin:
        out["point_list"] = _slice(binning, v.point_list, num_rendered, torch.int32)
        out["tile_ids"] = _slice(binning, v.tile_ids, num_rendered, torch.int32)
    return out
