#!/usr/bin/env python
"""Per-wave cycle trace of the blend backward (B3GS_BWD_TRACE=1): is the kernel bound by
throughput (all waves long) or by its longest serial chains (tail)?"""
import ctypes as C
import os
import sys
WHICH = sys.argv[1] if len(sys.argv) > 1 else "bwd"
os.environ["B3GS_BWD_TRACE" if WHICH == "bwd" else "B3GS_FWD_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from binocular3dgs_amd import _lib, synth
from binocular3dgs_amd.render import PipelineParams, render
P, W, H = 1_000_000, 800, 600
model = synth.synth_model(P, seed=0, device="cuda", width=W, height=H)
cam = synth.synth_cameras(W, H, yaws=(0.0,), device="cuda")[0]
bg = torch.zeros(3, device="cuda")
gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0, device="cuda")
for _ in range(3):
    pkg = render(cam, model, PipelineParams(), bg)
    if WHICH == "bwd":
        torch.autograd.backward([pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"]], [gc, gd, ga])
torch.cuda.synchronize()
L = _lib.lib()
L.b3gs_debug_bwd_trace.restype = C.c_size_t
L.b3gs_debug_bwd_trace.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros(1 << 20, np.uint64)
n = L.b3gs_debug_bwd_trace(buf.ctypes.data, buf.size)
t = buf[:n].reshape(-1, 4)
t = t[t[:, 0] > 0]
cyc = t[:, 0].astype(np.int64)
rs = (t[:, 1] >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF
re = (t[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
it, live = t[:, 2].astype(np.int64), (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
lanes = (t[:, 3] >> np.uint64(32)).astype(np.int64)
t0 = rs.min()
start_us, end_us = (rs - t0) / 100.0, (re - t0) / 100.0
dur_us = end_us - start_us
print("waves", len(t), "kernel span %.1f us" % end_us.max(), "last wave start %.1f us" % start_us.max())
print("shader clock (cycles / wall): median %.2f GHz" % np.median(cyc / np.maximum(dur_us, 1e-3) / 1e3))
for name, a in (("start_us", start_us), ("duration_us", dur_us), ("end_us", end_us), ("iters", it), ("live", live)):
    print(name, "mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (a.mean(), *np.percentile(a, [50, 90, 99]), a.max()))
print("cycles per iteration per wave: p50 %.0f" % np.median(cyc / np.maximum(it, 1)))
print("fraction of waves finished by 50%% / 75%% / 90%% of span: %.2f %.2f %.2f" % tuple((end_us < f * end_us.max()).mean() for f in (0.5, 0.75, 0.9)))
print("fraction of waves started after 10 us: %.2f" % (start_us > 10).mean())
if WHICH == "bwd":
    print("live lanes per live iteration: mean %.1f of 64 (%.0f%% of the evaluated lanes)" % (lanes.sum() / max(live.sum(), 1), 100.0 * lanes.sum() / max(live.sum(), 1) / 64))
print("corr(iters, duration) = %.3f" % np.corrcoef(it, dur_us)[0, 1])
