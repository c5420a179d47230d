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
    # empty tiles are stored as (0xFFFFFFFF, 0) (min / max identity of the tile sort's last pass): show [0, 0)
    out["ranges2"] = _slice(img, v.ranges2, tiles * 2, torch.int32).view(tiles, 2)
    for key in ("ranges", "ranges2"):
        empty = out[key][:, 1].to(torch.int64) <= (out[key][:, 0].to(torch.int64) & 0xFFFFFFFF)
        out[key] = torch.where(empty[:, None], torch.zeros_like(out[key]), out[key])
    out["counts"] = _slice(geom, v.counts, 3, torch.int32)      # N1, V, N2 (device side)
    if has_bin:
        words = _slice(binning, v.point_list, num_rendered, torch.int32)
        if v.packed_idx_bits >= 0:
            # one word per instance: (tile << bits) | index (b3gs_raster.h, B3gsDebugViews)
            w64 = words.to(torch.int64) & 0xFFFFFFFF
            out["point_list"] = (w64 & ((1 << v.packed_idx_bits) - 1)).to(torch.int32)
            out["tile_ids"] = (w64 >> v.packed_idx_bits).to(torch.int32)
            # segment 2 of a two-round forward sits behind segment 1 in the same array (N1 = counts[0], N2 = counts[2])
            out["point_list2"], out["tile_ids2"] = out["point_list"], out["tile_ids"]
        else:
            out["point_list"] = words
            out["tile_ids"] = _slice(binning, v.tile_ids, num_rendered, torch.int32)
            out["point_list2"], out["tile_ids2"] = out["point_list"], out["tile_ids"]
    return out
