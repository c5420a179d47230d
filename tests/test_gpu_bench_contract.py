"""GPU: `bench.py` keeps the driver's contract -- one JSON line, the named keys, the roofline / cpu_baseline objects --
on a small workload (so the default-size run cannot be the first time a key error shows up)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--gaussians", "30000",
           "--width", "208", "--height", "144", *extra]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    return json.loads(lines[-1])           # the JSON object is the LAST line of stdout


def test_default_line_has_the_contract_keys():
    d = _run("--no-extras", "--no-pmc")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "iters/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "bytes_per_launch", "avg_launch_ms"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "render_bwd_kernel"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] is None        # --no-pmc
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9) < 0.02 * r["achieved"] + 0.1
    # 44 B per processed tile instance + 28 B per pixel and view
    model = 44.0 * r["instances_per_launch"] + 28.0 * 208 * 144 * r["views_per_launch"]   # (the instance count is a mean over K steps)
    assert abs(r["bytes_per_launch"] - model) <= 44.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "iters/s" and c["cores"] >= 1 and c["value"] > 0 and "cpu_model" in c


def test_strong_scaling_and_dropin_flags_run():
    d = _run("--no-extras", "--no-pmc", "--no-cpu-baseline", "--scaling", "strong", "--views", "8")
    assert d["scaling"] == "strong" and d["config"]["global_views"] == 8 and d["config"]["views_per_rank"] == 8
    d = _run("--no-extras", "--no-pmc", "--no-cpu-baseline", "--path", "dropin", "--optimizer", "b3gs")
    assert d["config"]["path"] == "dropin" and d["config"]["hip_graph"] is False and d["value"] > 0
