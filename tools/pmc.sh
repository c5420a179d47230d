#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh <tag> "<COUNTER ...>" [kernel-regex]
# one rocprofv3 --pmc pass (counters only: never combined with trace domains other than
# kernel-trace) over tools/one_view.py -- or over $PMC_TARGET (e.g. "bench.py --steps 5 --warmup 2
# --no-cpu-baseline": the product path, batched launches); prints per-kernel averages of each counter
tag=$1; counters=$2; kre=${3:-render|preprocess|radix|emit|scan|tile_ranges}
ROOTDIR=$(pwd)
mkdir -p $ROOTDIR/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d /tmp/pmc_$tag -- python $ROOTDIR/${PMC_TARGET:-tools/one_view.py} > /tmp/pmc_$tag.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 /tmp/pmc_$tag.log; find /tmp/pmc_$tag | head; exit 1; fi
cp $f $ROOTDIR/gpurun_out/${tag}_counters.csv
cd $ROOTDIR
python - "$f" "$kre" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
kre = re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].split("<")[0][:40]
    if not kre.search(name): continue
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
