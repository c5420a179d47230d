"""usage (GPU box): tools/build_variant.sh bintrace "-DB3GS_BIN_TRACE" binning.hip; B3GS_LIB=tools/ab/bintrace.so python tools/bin_trace.py
Per-workgroup wall-clock stamps (100 MHz) of scan_chunk_sums and emit_instances of the LAST forward of the headline workload:
launch span, per-phase durations of the workgroups by kind."""
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
buf = np.zeros((2, 8192, 8), dtype=np.uint64)
L.b3gs_debug_bin_trace.restype = C.c_size_t
L.b3gs_debug_bin_trace(buf.ctypes.data_as(C.c_void_p))
us = lambda t: t.astype(np.float64) / 100.0
for k, name in ((0, "scan_chunk_sums"), (1, "emit_instances")):
    t = buf[k]
    live = t[:, 0] > 0
    t = t[live]
    t = t[t[:, 0] + 100 * 1000 > t[:, 0].max()]       # the last launch only (earlier, larger grids leave older stamps)
    t0 = t[:, 0].min()
    done = t[:, 5] > 0
    print(f"== {name}: {live.sum()} workgroups stamped, {done.sum()} ran to the end; first start -> last end "
          f"{us(t[done, 5].max() - t0):.1f} us; last START at {us(t[:, 0].max() - t0):.1f} us")
    w = t[done]
    dur = us(w[:, 5] - w[:, 0])
    print(f"   duration of a workgroup: mean {dur.mean():.1f} us, p50 {np.median(dur):.1f}, p95 {np.percentile(dur, 95):.1f}, max {dur.max():.1f}")
    if k == 0:
        c = w[w[:, 3] > 0]          # compacting workgroups
        d = w[w[:, 3] == 0]
        print(f"   dense workgroups {len(d)}: mean {us(d[:, 5] - d[:, 0]).mean():.1f} us, start at {us(d[:, 0] - t0).mean():.1f}")
        for a, b, what in ((0, 1, "bitmap staging"), (1, 2, "order + flag gathers + ballots"), (2, 3, "prefix + list"),
                           (3, 4, "list walk (rect gather, open tiles, stores)"), (4, 5, "block scans + sums")):
            x = us(c[:, b] - c[:, a])
            print(f"   compacting ({len(c)}): {what}: mean {x.mean():.1f} us, p95 {np.percentile(x, 95):.1f}, max {x.max():.1f}")
        print(f"   list entries per tile: mean {c[:, 7].astype(float).mean():.0f}, max {c[:, 7].max()}")
        print(f"   start of compacting workgroups after the first: mean {us(c[:, 0] - t0).mean():.1f} us, max {us(c[:, 0] - t0).max():.1f}")
    else:
        for kind, what in ((0, "dense (segment 1)"), (1, "compact (flagged lists)")):
            x = w[w[:, 7] == kind]
            if len(x):
                dd = us(x[:, 5] - x[:, 0])
                print(f"   {what}: {len(x)} workgroups, mean {dd.mean():.1f} us, p95 {np.percentile(dd, 95):.1f}, max {dd.max():.1f}; "
                      f"start mean {us(x[:, 0] - t0).mean():.1f}, end max {us(x[:, 5] - t0).max():.1f}")
