"""usage (GPU box): tools/build_variant.sh pretrace "-DB3GS_PRE_TRACE" preprocess.hip; B3GS_LIB=tools/ab/pretrace.so python tools/pre_trace.py
Per-workgroup wall-clock stamps (100 MHz) of the LAST preprocess_fwd_kernel launch of the headline workload."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda", 0)
job = bench.Job(args, dev, 0, 1, False, args.gaussians, args.width, args.height, args.fov, 6, "weak")
for _ in range(6):
    job.eager_step()
torch.cuda.synchronize()
L = __import__("binocular3dgs_amd._lib", fromlist=["lib"]).lib()
buf = np.zeros((8192, 12), dtype=np.uint64)
L.b3gs_debug_pre_trace.restype = C.c_size_t
L.b3gs_debug_pre_trace(buf.ctypes.data_as(C.c_void_p))
t = buf[buf[:, 0] > 0]
t = t[t[:, 0] + 100 * 1000 > t[:, 0].max()]
us = lambda x: x.astype(np.float64) / 100.0
t0 = t[:, 0].min()
print(f"{len(t)} workgroups; first start -> last end {us(t[:, 9].max() - t0):.1f} us")
dur = us(t[:, 9] - t[:, 0])
print(f"workgroup duration: mean {dur.mean():.1f} us, p5 {np.percentile(dur, 5):.1f}, p50 {np.median(dur):.1f}, p95 {np.percentile(dur, 95):.1f}, max {dur.max():.1f}")
print(f"prologue (parameter loads, covariance): mean {us(t[:, 1] - t[:, 0]).mean():.2f} us")
for v in range(6):
    nxt = t[:, 3 + v] if v < 5 else t[:, 9]
    print(f"view {v}: mean {us(nxt - t[:, 2 + v]).mean():.2f} us, p95 {np.percentile(us(nxt - t[:, 2 + v]), 95):.2f}")
# concurrency: workgroups in flight over time
ev = np.concatenate([np.stack([us(t[:, 0] - t0), np.ones(len(t))], 1), np.stack([us(t[:, 9] - t0), -np.ones(len(t))], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
span = us(t[:, 9].max() - t0)
for lo in np.arange(0, span, span / 10):
    m = (ev[:, 0] >= lo) & (ev[:, 0] < lo + span / 10)
    print(f"  t {lo:6.1f} us: workgroups in flight mean {conc[m].mean():7.1f}" if m.any() else "")
hw = t[:, 10].astype(np.int64)
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = t[:, 11].astype(np.int64) & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct (xcc, se, sh, cu):", len(np.unique(key)), "workgroups per CU: mean", len(t) / len(np.unique(key)))
starts = us(t[:, 0] - t0)
print("start times: p25 %.1f p50 %.1f p75 %.1f max %.1f" % tuple(np.percentile(starts, [25, 50, 75, 100])))
