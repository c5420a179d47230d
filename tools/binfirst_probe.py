#!/usr/bin/env python
"""VERDICT r4 item 4, the estimate BEFORE building: "bin first, order later" -- emit the tile instances unordered into
per-tile segments (atomic cursors), then sort every tile's list by (depth bits, Gaussian index) in LDS -- timed on the
headline's REAL instance stream (1M Gaussians, 800x600, 6 views in one launch) with the cheapest honest form of each stage
(tools/ubench/binfirst.hip), against the chain it would replace: depth sort of the P Gaussians + scan + in-order emission +
two tile-split radix passes (per-kernel times of the same iteration: profiles/r04_final_default_kernel_stats.csv).
Two instance streams: what one-round tight binning emits (4.8M per view) and what two-round binning hands the tile split
(segment 1: ~1.5M per view; segment 1 is defined BY the depth order, so a bin-first design would need a depth threshold
predicted from the previous iteration in its place -- that cost is not included here)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from binocular3dgs_amd import synth  # noqa: E402
from binocular3dgs_amd.debug import state_views  # noqa: E402
from binocular3dgs_amd.fused import FusedRasterizer  # noqa: E402

P, W, H = 1_000_000, 800, 600
dev = "cuda"
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "binfirst.so"))
lib.binfirst_run.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_uint32] + [C.c_void_p] * 7 + [C.c_int, C.POINTER(C.c_float)]
lib.binfirst_run.restype = C.c_int
model = synth.synth_model(P, seed=0, device=dev, width=W, height=H)
pairs = synth.synth_view_set(W, H, device=dev)
bg = torch.zeros(3, device=dev)
views = []
for i, (c, s, t) in enumerate(pairs):
    views += [(c, 2 * i), (s, 2 * i + 1)]
tiles_per_view = ((W + 15) // 16) * ((H + 15) // 16)


def stream(seg1):
    fr = FusedRasterizer(model, W, H, num_slots=6, want_means2D=False, seg1_fraction=seg1)
    fr.fit_capacity(views, bg)
    with torch.no_grad():
        for _ in range(4):                       # (two rounds: the open-tile prediction settles)
            fr.render_batch([(c, s, False) for c, s in views], bg)
    torch.cuda.synchronize()
    tl, ky, ix = [], [], []
    for k, (_, s) in enumerate(views):
        sl = fr.slots[s]
        v = state_views(P, W, H, sl.capacity, sl.geom, sl.binning, sl.img)
        n1 = int(v["counts"][0])
        pl = v["point_list"][:n1].long()
        td = v["tile_ids"][:n1].long()
        o = torch.argsort(pl * 4096 + td)        # emission order of an UNSORTED emit: by Gaussian, its tiles consecutive
        tl.append((td[o] + k * tiles_per_view).int())
        ky.append(v["depth_bits"].long()[pl[o]].int())
        ix.append(pl[o].int())
    return torch.cat(tl).contiguous(), torch.cat(ky).contiguous(), torch.cat(ix).contiguous(), fr.seg1_fraction


for name, seg1 in (("one round (every tight-binned instance)", 0.0), ("two rounds (segment 1 + predicted-open tiles)", "auto")):
    tile, key, idx, frac = stream(seg1)
    n, tiles = tile.numel(), 6 * tiles_per_view
    cnt = torch.bincount(tile.long(), minlength=tiles)
    begin = torch.zeros(tiles + 1, dtype=torch.int64, device=dev)
    begin[1:] = torch.cumsum(cnt, 0)
    begin32 = begin.int().contiguous()
    order = torch.argsort(cnt, descending=True).int().contiguous()
    w_cnt, w_cur = torch.zeros(tiles, dtype=torch.int32, device=dev), torch.zeros(tiles, dtype=torch.int32, device=dev)
    okey, oidx = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    times = (C.c_float * 5)()
    torch.cuda.synchronize()
    rc = lib.binfirst_run(tile.data_ptr(), key.data_ptr(), idx.data_ptr(), n, tiles, begin32.data_ptr(), order.data_ptr(),
                          w_cnt.data_ptr(), w_cur.data_ptr(), okey.data_ptr(), oidx.data_ptr(), out.data_ptr(), 10, times)
    torch.cuda.synchronize()
    assert rc == 0
    # the result is what the blend kernels need: every tile's pairs ascending by (depth key, index)
    k64 = (out[:, 0].long() & 0xFFFFFFFF) * (1 << 21) + out[:, 1].long()
    tile_of = torch.repeat_interleave(torch.arange(tiles, device=dev), cnt)
    ok = bool(((k64[1:] >= k64[:-1]) | (tile_of[1:] != tile_of[:-1])).all()) if int(cnt.max()) <= 4096 else None
    long_tiles = int((cnt > 2048).sum()), int((cnt > 4096).sum())
    print(f"{name}: seg1_fraction={frac}  instances per launch (6 views) {n} = {n / 6e6:.2f}M per view; tiles {tiles}, "
          f"list length mean {n / tiles:.0f} p50 {int(cnt.float().median())} max {int(cnt.max())}; tiles > 2048: {long_tiles[0]}, > 4096: {long_tiles[1]}")
    print(f"   count {times[0]:.1f} us | scatter 2x4B {times[1]:.1f} us | scatter 8B {times[2]:.1f} us | "
          f"tile sort CAP 2048 {times[3]:.1f} us | CAP 4096 {times[4]:.1f} us | lists sorted: {ok}")
    best = times[0] + min(times[1], times[2]) + min(times[3], times[4])
    print(f"   bin-first chain >= {best:.1f} us per iteration (+ ~5 us of launch floor per stage, + the per-tile count inside "
          f"the projection, + a scan over {tiles} tiles)")
print("replaced chain at the headline (profiles/r04_final_default_kernel_stats.csv, two rounds): depth sort 129 + scan 47 + "
      "emission 36 + tile split 129 = ~341 us per iteration; kill criterion: the new chain must be >= 100 us below it")
