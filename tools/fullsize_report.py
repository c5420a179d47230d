#!/usr/bin/env python
"""GPU box: print (and write to gpurun_out/fullsize_report.json) the HIP-vs-oracle metrics of tests/fullsize.py at
BASELINE.json's sizes.  usage: python tools/fullsize_report.py [small]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fullsize  # noqa: E402

cases = [("500k", 500_000, 800, 600, 60.0, 6), ("1M", 1_000_000, 800, 600, 60.0, 6),
         ("2M_1600", 2_000_000, 1600, 1600, 50.0, 8)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    cases = [("20k", 20_000, 320, 240, 60.0, 6)]
out = {}
for name, P, W, H, fov, views in cases:
    t0 = time.time()
    out[name + "_dropin"] = fullsize.dropin_metrics(P, W, H, fov)
    t1 = time.time()
    out[name + "_fused"] = fullsize.fused_metrics(P, W, H, fov, views=views)
    out[name + "_seconds"] = [round(t1 - t0, 1), round(time.time() - t1, 1)]
    print(name, json.dumps({k: out[k] for k in out if k.startswith(name)}), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fullsize_report.json"), "w"), indent=1)
