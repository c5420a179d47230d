#!/usr/bin/env python
"""Per-wave trace of the BATCHED blend kernels (all 6 views of an iteration in one launch; B3GS_BWD_TRACE / B3GS_FWD_TRACE):
how many waves are resident over the kernel's span -- i.e. how much of the launch is tail."""
import ctypes as C
import os
import sys
WHICH = sys.argv[1] if len(sys.argv) > 1 else "bwd"
os.environ["B3GS_BWD_TRACE" if WHICH == "bwd" else "B3GS_FWD_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from binocular3dgs_amd import _lib, synth
from binocular3dgs_amd.fused import FusedRasterizer
P, W, H = (int(x) for x in os.environ.get("B3GS_TRACE_SHAPE", "1000000,800,600").split(","))
model = synth.synth_model(P, seed=0, device="cuda", width=W, height=H)
pairs = synth.synth_view_set(W, H, device="cuda")
bg = torch.zeros(3, device="cuda")
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0, device="cuda")
fr = FusedRasterizer(model, W, H, num_slots=6, want_means2D=False)
views = []
for i, (c, s, t) in enumerate(pairs):
    views += [(c, 2 * i, True), (s, 2 * i + 1, False)]
fr.fit_capacity(views, bg)
for p in model.parameters():
    p.grad = torch.zeros_like(p)
for _ in range(3):
    outs = fr.render_batch(views, bg)
    if WHICH == "bwd":
        o, g = [], []
        for k, x in enumerate(outs):
            o.append(x["render"]); g.append(gc)
            if k % 2 == 0:
                o += [x["rendered_depth"], x["rendered_alpha"]]; g += [gd, ga]
        torch.autograd.backward(o, g)
torch.cuda.synchronize()
L = _lib.lib()
L.b3gs_debug_bwd_trace.restype = C.c_size_t
L.b3gs_debug_bwd_trace.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros(1 << 22, np.uint64)
n = L.b3gs_debug_bwd_trace(buf.ctypes.data, buf.size)
t = buf[:n].reshape(-1, 4)
t = t[t[:, 0] > 0]
rs = (t[:, 1] >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF
re = (t[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
it = (t[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
n_half = ((t[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64)
n_pair = ((t[:, 2] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.float64)
t0 = rs.min()
s_us, e_us = (rs - t0) / 100.0, (re - t0) / 100.0
span = e_us.max()
print(WHICH, "waves", len(t), "kernel span %.1f us" % span, "last wave start %.1f us" % s_us.max())
grid = np.linspace(0, span, 21)
res = [int(((s_us <= x) & (e_us > x)).sum()) for x in grid]
print("resident waves at 0,5,...,100 %% of the span:", res)
print("mean resident waves %.0f (slots: 1024 SIMDs x occupancy)" % ((e_us - s_us).sum() / span))
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % ((e_us - s_us).mean(), *np.percentile(e_us - s_us, [50, 90, 99]), (e_us - s_us).max()))
print("iterations per wave: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (it.mean(), *np.percentile(it, [50, 90, 99]), it.max()))
print("corr(iters, duration) %.3f" % np.corrcoef(it, e_us - s_us)[0, 1])
if WHICH == "bwd":
    # candidates that reached the reduction (any pixel of the quadrant live) and how full their lanes were
    nlive = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64)
    nlanes = (t[:, 3] >> np.uint64(32)).astype(np.float64)
    print("candidates evaluated %.3e, with a live lane %.3e (%.1f %%); live lanes per live candidate %.1f of 64 (%.1f %%)"
          % (it.sum(), nlive.sum(), 100.0 * nlive.sum() / max(it.sum(), 1), nlanes.sum() / max(nlive.sum(), 1),
             100.0 * nlanes.sum() / max(64.0 * nlive.sum(), 1)))
    print("live candidates confined to one half of the quadrant (rows 0-3 / 4-7 / columns 0-3 / 4-7): %.1f %%; adjacent pairs "
          "in complementary halves (what one loop trip could serve together): %.1f %% of the live candidates"
          % (100.0 * n_half.sum() / max(nlive.sum(), 1), 100.0 * 2.0 * n_pair.sum() / max(nlive.sum(), 1)))
    frac = nlanes / np.maximum(64.0 * nlive, 1)
    w = nlive > 0
    print("per-wave live-lane fraction: p10 %.2f p50 %.2f p90 %.2f; waves below 1/2: %.1f %% holding %.1f %% of the live candidates"
          % (*np.percentile(frac[w], [10, 50, 90]), 100.0 * (frac[w] < 0.5).mean(), 100.0 * nlive[w][frac[w] < 0.5].sum() / nlive.sum()))
late = s_us > 0.5 * span
print("waves started in the second half: %d, their mean duration %.1f us" % (late.sum(), (e_us - s_us)[late].mean() if late.any() else 0))
# imbalance between the four quadrant waves of a workgroup (they live until the tile is done): the share of wave-slot time
# held by waves that had fewer candidates than the busiest wave of their tile
full = buf[:n].reshape(-1, 4)
if len(full) % 4 == 0:
    wg_it = (full[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.float64).reshape(-1, 4)
    wg_dur = ((full[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64) - ((full[:, 1] >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF)).astype(np.float64).reshape(-1, 4) / 100.0
    live = wg_it.max(1) > 0
    mx = wg_it[live].max(1, keepdims=True)
    idle = (wg_dur[live] * (1.0 - wg_it[live] / mx)).sum() / wg_dur[live].sum()
    print("workgroups %d; candidates per wave / busiest wave of its tile: mean %.2f; wave-slot time held by the lighter waves: %.1f %%"
          % (int(live.sum()), float((wg_it[live] / mx).mean()), 100.0 * idle))
