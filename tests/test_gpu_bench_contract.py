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
    assert r["bound"] == "valu" and "HBM" in r["bound_note"] and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "render_bwd_kernel"
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


@pytest.mark.parametrize("world,dp_extras", [(2, 1), (8, 1), (2, 0)])
def test_ranks_sharing_the_gpu_run_the_multi_gpu_control_flow(world, dp_extras):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` as the driver launches it, except that
    all ranks use cuda:0 and gloo carries the collectives (test hooks of bench.py): weak-scaling headline, the
    strong-scaling extra (6 views over 2 ranks: one split pair) and the 8-view extra, reduce-scatter -> sharded Adam ->
    all-gather, max-over-ranks timing, ONE JSON line from rank 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, B3GS_BENCH_BACKEND="gloo", B3GS_BENCH_SINGLE_DEVICE="1", B3GS_BENCH_SMALL_EXTRAS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--gaussians", "30000", "--width", "208", "--height", "144", "--dp-extras", str(dp_extras)]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["global_views"] == 6 * world
    assert d["config"]["views_per_rank"] == 6 and d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["traffic"] is None
    x = d["exchange"]          # what the SCALE record is checked against: the group, every rank's views, the collectives' time
    assert x["backend"] == "gloo" and x["rccl_ranks"] == world and x["views_per_rank"] == [6] * world
    # N > 1 default: replicated one-launch Adam behind the range-pipelined all-reduce tail; the number of ranges follows P
    # (step.auto_pipeline_ranges: ONE at this test's 30k Gaussians -- still the in-band overflow word, no agreement collective)
    assert x["optimizer"] == "FusedAdam" and x["ranges"] == 1 and d["config"]["dp_tail_ranges"] == 1
    assert x["tail_ms"] > 0 and len(x["all_reduce_window_ms"]) == 1 and x["exchange_bytes_per_rank"] > 0
    assert "weak scaling" in d["metric"] and f"{world} x 6 views" in d["metric"]
    if not dp_extras:      # opt-out: the headline only
        assert "extras" not in d and "extras_skipped" in d
        return
    ex = d["extras"]
    # 6 views over 2 ranks: 3 + 3 (one split pair); over 8 ranks: one view each, two ranks idle; 8 views: 4 + 4 / one each
    assert ex["strong_scaling_6_views"]["views_per_rank"] == ([3, 3] if world == 2 else [1, 1, 1, 1, 1, 1, 0, 0])
    assert ex["strong_scaling_6_views"]["iters_per_s"] > 0
    assert ex["config5_2M_1600x1600_8_views"]["views_per_rank"] == ([4, 4] if world == 2 else [1] * 8)


@pytest.mark.parametrize("fault", ["raise:1", "hang:1"])
def test_a_rank_that_fails_or_hangs_in_a_strong_scaling_leg_does_not_take_the_headline_down(fault):
    """The strong-scaling legs are part of the DEFAULT N > 1 line (round 6).  A rank that raises while building its leg: the
    status word every rank all-reduces after each phase makes ALL ranks drop that leg (`error` in its place), the next leg and
    the headline are unaffected.  A rank that hangs: the deadline prints the headline line with what was measured until then
    and every rank exits with code 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, B3GS_BENCH_BACKEND="gloo", B3GS_BENCH_SINGLE_DEVICE="1", B3GS_BENCH_SMALL_EXTRAS="1",
               B3GS_BENCH_TEST_FAULT=fault)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--gaussians", "30000", "--width", "208", "--height", "144", "--dp-extras-deadline", "25"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["exchange"]["rccl_ranks"] == 2
    ex = d["extras"]
    if fault.startswith("raise"):
        assert "error" in ex["strong_scaling_6_views"] and "iters_per_s" not in ex["strong_scaling_6_views"]
        assert ex["config5_2M_1600x1600_8_views"]["views_per_rank"] == [4, 4]
    else:
        assert "deadline" in ex and "strong_scaling_6_views" not in ex


def test_sharded_adam_stays_selectable_for_n_ranks():
    """--optimizer sharded at N > 1: reduce-scatter -> Adam on 1/N -> all-gather, with the chain rule behind the overflow
    agreement (two ranks on cuda:0 over gloo)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, B3GS_BENCH_BACKEND="gloo", B3GS_BENCH_SINGLE_DEVICE="1", B3GS_BENCH_SMALL_EXTRAS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--gaussians", "30000", "--width", "208", "--height", "144", "--no-extras", "--optimizer", "sharded"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])
    x = d["exchange"]
    assert x["optimizer"] == "ShardedAdam" and x["reduce_scatter_ms"] > 0 and x["all_gather_ms"] > 0


def test_pipelined_tail_on_a_one_rank_nccl_group():
    """The N > 1 default (--optimizer b3gs, four ranges) on the REAL backend: async all_reduce per range + wait(), issued
    with the arguments an 8-GPU run issues them with; the exchange record carries the overlapped schedule."""
    d = _run("--dp-path", "--no-extras", "--no-pmc", "--no-cpu-baseline", "--optimizer", "b3gs", "--pipeline-ranges", "4")
    x = d["exchange"]
    assert x["backend"] == "nccl" and x["rccl_ranks"] == 1 and x["optimizer"] == "FusedAdam" and x["ranges"] == 4
    assert d["config"]["dp_tail_ranges"] == 4 and x["tail_ms"] > 0 and x["chain_rule_ms"] > 0
    assert len(x["all_reduce_window_ms"]) == 4 and x["all_reduce_span_ms"] > 0


@pytest.mark.parametrize("extra", [pytest.param(("--scaling", "strong", "--views", "8"), id="strong_8_views"),
                                   pytest.param(("--dp-graph", "1"), id="weak_collectives_in_graph")])
def test_rccl_call_signatures_on_a_one_rank_nccl_group(extra):
    """`bench.py --dp-path`: the N > 1 code path on ONE MI355X with the REAL backend -- a 1-rank "nccl" (= RCCL) process
    group, so that reduce_scatter_tensor / all_gather_into_tensor / all_reduce / barrier are issued with exactly the
    arguments an 8-GPU run issues them with (every other N > 1 test uses gloo).  strong_8_views: configs[4]'s view-granular
    sharding; weak_collectives_in_graph: the collectives and the Adam launch captured in the iteration's HIP graph."""
    d = _run("--dp-path", "--no-extras", "--no-pmc", "--no-cpu-baseline", "--optimizer", "sharded", *extra)
    assert d["value"] > 0 and d["n_gpus"] == 1
    x = d["exchange"]
    assert x["backend"] == "nccl" and x["rccl_ranks"] == 1 and x["optimizer"] == "ShardedAdam"
    assert x["reduce_scatter_ms"] > 0 and x["all_gather_ms"] > 0
    assert x["views_per_rank"] == [8 if "--views" in extra else 6]
    if "--dp-graph" in extra:
        assert x["dp_graph"] is True, "the RCCL collectives must be capturable in the HIP graph"


def test_gpus_n_without_a_launcher_spawns_n_ranks_itself():
    """`python bench.py --gpus 2` (the form the driver uses for N = 1) must not silently measure one rank: with no
    WORLD_SIZE around it the script becomes the launcher (torch.distributed.run, one process per GPU -- here both ranks on
    cuda:0 over gloo, the test hooks); a launcher whose rank count differs from --gpus is refused."""
    env = dict(os.environ, B3GS_BENCH_BACKEND="gloo", B3GS_BENCH_SINGLE_DEVICE="1", B3GS_BENCH_SMALL_EXTRAS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--gaussians", "30000",
           "--width", "208", "--height", "144", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["exchange"]["rccl_ranks"] == 2 and d["config"]["global_views"] == 12
    # a launcher that started ONE rank for --gpus 2
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env1)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
