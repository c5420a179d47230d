"""G11 (tests/golden/train_trace.json, recorded from the reference's training loop by tests/golden/make_golden_trace.py):
the build's own driver of one training iteration (binocular3dgs_amd/schedule.py) must issue the SAME calls -- order,
argument shapes / dtypes / constants, which output feeds which input, means of derived arguments (the disparity expression),
the weights of the loss terms at backward(), the masked write into max_radii2D -- when it is handed the same stand-ins."""
import json
import os
import types

import pytest
import torch

from binocular3dgs_amd.schedule import IterationSchedule
from trace_recorder import Flow, ModelStandIn, Recorder, SceneStandIn, make_callables, strip_optional

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_trace.json")


def _draws(golden_iterations):
    """The view and the baseline the reference's RNG drew in each recorded iteration (read off its render / shift calls)."""
    out = []
    for it in golden_iterations:
        view, shift = None, None
        for e in it["events"]:
            if e["call"] == "render#0":
                view = e["args"]["camera"]["camera"]
            if e["call"] == "getShiftedCamera#0":
                shift = e["args"]["trans_dist"]
        out.append((it["iteration"], view, shift))
    return out


def _drive(run, log_item):
    flags = run["flags"]
    rec = Recorder()
    ops = make_callables(rec)
    model = ModelStandIn(rec)
    scene = SceneStandIn(rec, model)
    model.training_setup(None)
    o = flags["opt"]
    pipe = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    with Flow(rec):
        sched = IterationSchedule(model, scene, pipe, torch.zeros(3), ops=ops, iterations=o["iterations"],
                                  shift_cam_start=flags["shift_cam_start"], binocular=flags["args"]["binocular_consistency"],
                                  opacity_decay_factor=flags["opacity_decay_factor"] if flags["args"]["opacity_decay"] else None,
                                  lambda_dssim=o["lambda_dssim"], densify_from_iter=o["densify_from_iter"],
                                  densify_until_iter=o["densify_until_iter"], densification_interval=o["densification_interval"],
                                  densify_grad_threshold=o["densify_grad_threshold"], log_item=log_item)
        for it, view, shift in _draws(run["iterations"]):
            sched.run_iteration(it, view, shift)
    return rec


def _first_difference(a, b, path=""):
    if type(a) is not type(b):
        return f"{path}: {a!r} != {b!r}"
    if isinstance(a, dict):
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b:
                return f"{path}.{k}: only on one side"
            d = _first_difference(a[k], b[k], f"{path}.{k}")
            if d:
                return d
        return None
    if isinstance(a, list):
        if len(a) != len(b):
            return f"{path}: {len(a)} entries != {len(b)}"
        for i, (x, y) in enumerate(zip(a, b)):
            d = _first_difference(x, y, f"{path}[{i}]")
            if d:
                return d
        return None
    if isinstance(a, float):
        return None if abs(a - b) <= 1e-6 * max(1.0, abs(a)) else f"{path}: {a} != {b}"
    return None if a == b else f"{path}: {a!r} != {b!r}"


@pytest.mark.parametrize("run_name", ["default", "plain"])
@pytest.mark.parametrize("log_item", [False, True])
def test_driver_reproduces_the_recorded_call_sequence(run_name, log_item):
    golden = json.load(open(GOLDEN))
    run = golden[run_name]
    rec = _drive(run, log_item)
    want = strip_optional(run["iterations"], keep_item=log_item)
    got = strip_optional(json.loads(json.dumps(rec.iterations)), keep_item=log_item)
    assert [e["call"] for e in rec.setup] == [e["call"] for e in run["setup"]]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert [e["call"] for e in g["events"]] == [e["call"] for e in w["events"]], f"iteration {w['iteration']}"
        diff = _first_difference(w, g, f"it{w['iteration']}")
        assert diff is None, diff


def test_trace_covers_both_sides_of_every_branch():
    golden = json.load(open(GOLDEN))
    calls = {name: [[e["call"] for e in it["events"]] for it in golden[name]["iterations"]] for name in ("default", "plain")}
    d, p = calls["default"], calls["plain"]
    assert any("render#1" in c for c in d) and any("render#1" not in c for c in d)                  # shift_cam_start
    assert any("opacity_decay#0" in c for c in d) and any("opacity_decay#0" not in c for c in d)    # densify_from_iter
    assert sum("densify_and_prune#0" in c for c in d) == 2
    assert "optimizer.step#0" not in d[-1] and "optimizer.step#0" in d[0]                            # the last iteration
    assert any("setitem" in c for c in p) and any("setitem" not in c for c in p)                    # densify_until_iter
    assert not any("render#1" in c for c in p)
    # the weights the reference's loss sum hands to backward()
    ev = [e for e in golden["default"]["iterations"][6]["events"] if e["call"] == "backward"][0]
    assert ev["weights"] == {"l1_loss#0": 1.0, "l1_loss#1": 0.8, "smooth_loss#0": 0.05, "ssim#0": -0.2}
