#!/usr/bin/env python
"""usage (GPU box): python tools/ab_interleaved.py <experiment> [P W H] -- settings compared INTERLEAVED inside one process on the
warmed-up step of bench_ref_schedule.py's surfaces: blocks of 40 iterations cycle through the settings, so all of them see the
same host; median and best block per setting.  (Two runs of the same code differ by +-15 % on the shared host of the GPU box:
separate runs cannot resolve a 10 % effect.)  Experiments:
  lazy   the pending pair forward of render(): adaptive rule (default) | always wait for the partner | never (every view at once)
  lazymax  B3GS_DROPIN_LAZY_MAX 2 | 8"""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import bench_ref_schedule as b                      # noqa: E402
import binocular3dgs_amd.rasterizer as R            # noqa: E402

exp = sys.argv[1] if len(sys.argv) > 1 else "lazy"
P, W, H = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (500_000, 504, 378)
dev = torch.device("cuda:0")


def set_lazy(mode):
    R._flush_pending()
    R._LAZY_FWD, R._LAZY_WHEN_IDLE = {"adaptive": (True, False), "always": (True, True), "never": (False, False)}[mode]
    R._S.adapt.clear()


def set_lazy_max(mode):
    R._flush_pending()
    R._LAZY_FWD, R._LAZY_WHEN_IDLE, R._LAZY_ADAPT = True, True, False
    R._LAZY_MAX = int(mode)


EXPERIMENTS = {"lazy": (("adaptive", "always", "never"), set_lazy),
               # renders per pending forward: 2 = the second render of a pair launches the batch when it returns; 8 = nothing
               # launches until an output is touched (or eight renders are pending)
               "lazymax": (("2", "8"), set_lazy_max)}
modes, setter = EXPERIMENTS[exp]


def probe(step, rounds=12, n=40):
    res = {m: [] for m in modes}
    for _ in range(rounds):
        for m in modes:
            setter(m)
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize(dev)
            res[m].append(n / (time.perf_counter() - t0))
    setter(modes[0])
    return {m: (round(statistics.median(v), 1), round(max(v), 1)) for m, v in res.items()}


for surf in ("unchanged", "render"):
    r = b.run(dev, P, W, H, 60.0, surf, probe=probe)
    print(f"{surf:10s} P={P} {W}x{H}  " + " | ".join(f"{m}: median {v[0]} best {v[1]}" for m, v in r.items()))
