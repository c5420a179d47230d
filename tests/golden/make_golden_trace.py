#!/usr/bin/env python
"""G11: tests/golden/train_trace.json -- the per-iteration CALL SEQUENCE of the reference's training loop
(train.py:65-202), recorded by IMPORTING the reference's train module under the CPU shim of make_golden.py and running its
`training()` with the recording stand-ins of tests/trace_recorder.py in the place of everything it calls:

    render, l1_loss, ssim, SmoothLoss, inverse_warp_images          (train.py:17-25 imports)
    Scene (getTrainCameras, getShiftedCamera, cameras_extent), GaussianModel (training_setup, update_learning_rate,
    oneupSHdegree, opacity_decay, add_densification_stats, densify_and_prune, optimizer.step / zero_grad, max_radii2D)

Two short runs so that both sides of every per-iteration branch are seen:
    "default"   the reference's default flags (binocular consistency and opacity decay on): iterations either side of
                shift_cam_start, of densify_from_iter, two densifications, the last iteration (no optimizer step)
    "plain"     both flags off and densify_until_iter inside the run: iterations either side of densify_until_iter
What is stored is DATA: ordered call names, argument shapes / dtypes / scalar constants, which earlier output every tensor
argument derives from, the mean of derived arguments, the loss weights seen by backward(), `.item()` read-backs and masked
writes into model state.  No source text.  Re-run with:  python tests/golden/make_golden_trace.py"""
import json
import os
import random
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from make_golden import OUT, CudaToCpu, install_shim  # noqa: E402
from trace_recorder import Flow, ModelStandIn, Recorder, SceneStandIn, make_callables  # noqa: E402

RUNS = {
    "default": dict(args=dict(binocular_consistency=True, opacity_decay=True), opt=dict(densify_until_iter=8)),
    "plain": dict(args=dict(binocular_consistency=False, opacity_decay=False), opt=dict(densify_until_iter=7)),
}
ITERATIONS, SHIFT_CAM_START, DENSIFY_FROM, DENSIFY_EVERY = 12, 4, 2, 5


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self):
        pass

    def elapsed_time(self, other):
        return 0.0


class _Bar:
    def __init__(self, *a, **k):
        pass

    def set_postfix(self, *a, **k):
        pass

    def update(self, *a, **k):
        pass

    def close(self):
        pass


def record_run(train, flags):
    rec = Recorder()
    ns = make_callables(rec)
    made = {}

    def model_factory(sh_degree):
        made["model"] = ModelStandIn(rec, sh_degree)
        return made["model"]

    def scene_factory(dataset, gaussians):
        made["scene"] = SceneStandIn(rec, gaussians)
        return made["scene"]

    patch = dict(render=ns.render, l1_loss=ns.l1_loss, ssim=ns.ssim, SmoothLoss=ns.SmoothLoss,
                 inverse_warp_images=ns.inverse_warp_images, GaussianModel=model_factory, Scene=scene_factory,
                 prepare_output_and_logger=lambda d: None, training_report=lambda *a, **k: None, tqdm=_Bar)
    old = {k: getattr(train, k) for k in patch}
    old_event = torch.cuda.Event
    for k, v in patch.items():
        setattr(train, k, v)
    torch.cuda.Event = _Event
    try:
        dataset = types.SimpleNamespace(sh_degree=1, white_background=False, model_path="", source_path="synthetic")
        opt = types.SimpleNamespace(iterations=ITERATIONS, random_background=False, lambda_dssim=0.2,
                                    densify_from_iter=DENSIFY_FROM, densification_interval=DENSIFY_EVERY,
                                    opacity_reset_interval=3000, densify_grad_threshold=0.0002, **flags["opt"])
        pipe = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
        args = types.SimpleNamespace(start_checkpoint=None, debug_from=-1, dataset_name="LLFF", source_path="synthetic",
                                     shift_cam_start=SHIFT_CAM_START, cam_trans_dist=0.4, opacity_decay_factor=0.995,
                                     test_iterations=[], save_iterations=[], checkpoint_iterations=[], **flags["args"])
        random.seed(2024)
        torch.manual_seed(2024)
        with CudaToCpu(), Flow(rec):
            train.training(dataset, opt, pipe, args)
    finally:
        for k, v in old.items():
            setattr(train, k, v)
        torch.cuda.Event = old_event
    return {"flags": {"args": flags["args"], "opt": dict(flags["opt"], iterations=ITERATIONS, densify_from_iter=DENSIFY_FROM,
                                                        densification_interval=DENSIFY_EVERY, lambda_dssim=0.2,
                                                        densify_grad_threshold=0.0002),
                      "shift_cam_start": SHIFT_CAM_START, "cam_trans_dist": 0.4, "opacity_decay_factor": 0.995},
            "setup": rec.setup, "iterations": rec.iterations}


def main():
    install_shim({})
    torch.nn.Module.cuda = lambda self, *a, **k: self
    with CudaToCpu():
        import train                                              # /root/reference/train.py (read-only import)
    out = {"what": "call sequence of the reference's training loop, recorded with stand-ins (tests/trace_recorder.py)",
           "image": [Recorder().H, Recorder().W]}
    for name, flags in RUNS.items():
        out[name] = record_run(train, flags)
    path = os.path.join(OUT, "train_trace.json")
    with open(path, "w") as f:
        json.dump(out, f, sort_keys=True, separators=(",", ":"))
    n = sum(len(it["events"]) for r in RUNS for it in out[r]["iterations"])
    print(f"wrote {path}: {n} events")


if __name__ == "__main__":
    main()
