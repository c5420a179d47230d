#!/usr/bin/env python
"""Per-kernel VALU picture from two rocprofv3 --pmc passes (tools/pmc.sh; counters only + kernel-trace):
  pass A: SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES
  pass B: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
usage: tools/valu_summary.py <a_counters.csv> <b_counters.csv> <out.csv>
Columns (per-launch averages; counters are summed over the 8 XCDs by rocprofv3):
  cycles          = GRBM_GUI_ACTIVE / 8: shader-clock cycles the launch occupied the GPU
  lanes_per_inst  = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU: active lanes per wave64 VALU instruction (64 = no divergence)
  valu_issue_frac = SQ_INSTS_VALU * 2 / (cycles * 1024): share of the chip's VALU issue slots used, at the measured
                    2 cycles per plain fp32 wave64 instruction (tools/ubench/valu_rate.hip), 256 CUs x 4 SIMDs;
                    transcendental / DPP instructions occupy 8 cycles, so a kernel rich in them saturates below 1
  valu_lane_frac  = SQ_THREAD_CYCLES_VALU / (cycles * 256 CUs * 128 lanes): the same, counting only active lanes
"""
import collections
import csv
import re
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_n": len(next(iter(cs.values())))} for k, cs in acc.items()}


a, b = load(sys.argv[1]), load(sys.argv[2])
cols = ["kernel", "launches", "cycles", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU",
        "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES", "lanes_per_inst", "valu_issue_frac",
        "valu_lane_frac"]
rows = []
for k in a:
    if k not in b or not re.search(r"render|preprocess|radix|emit|scan|accumulate|adam|loss|ssim", k):
        continue
    cyc = b[k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    iv, tc = a[k].get("SQ_INSTS_VALU", 0.0), b[k].get("SQ_THREAD_CYCLES_VALU", 0.0)
    rows.append([k, a[k]["_n"], round(cyc), round(iv), round(a[k].get("SQ_INSTS_SALU", 0)), round(a[k].get("SQ_INSTS_LDS", 0)),
                 round(tc), round(b[k].get("SQ_ACTIVE_INST_VALU", 0)), round(b[k].get("SQ_WAIT_INST_ANY", 0)),
                 round(a[k].get("SQ_BUSY_CYCLES", 0)), round(a[k].get("SQ_WAVES", 0)),
                 round(tc / iv, 2) if iv else 0, round(iv * 2 / (cyc * 1024), 4) if cyc else 0,
                 round(tc / (cyc * 256 * 128), 4) if cyc else 0])
rows.sort(key=lambda r: -r[2] * r[1])
w = csv.writer(open(sys.argv[3], "w"))
w.writerow(cols)
w.writerows(rows)
for r in rows:
    print(dict(zip(cols, r)))
