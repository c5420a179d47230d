"""The reference's OWN per-iteration shape as a timed loop (bench.py extras `reference_schedule_*`).

train.py:92-198 per iteration: ONE randomly chosen input view (:92), render (:100), ONE randomly shifted partner built by
Scene.getShiftedCamera (:124-128: a host sync and a new Camera every iteration), render, the binocular loss block
(:128-148), backward, opacity decay BEFORE the optimiser step (:171-173), densification statistics (:178-179), Adam
(:196-198).  Two views per iteration and cameras that change every step: about forty kernel launches of a few
microseconds each, so launch latency -- not the blend kernels -- sets the pace, and the open-tile prediction of the
two-round binning cannot settle (the rule falls back to one round by itself when it keeps missing).

Three surfaces, same Gaussians, same camera / shift sequence, random ground-truth images (throughput only):
  "fused"     the build's own step: FusedRasterizer pair batch (shared depth sort) + fused loss block + one-launch Adam
              with the reference's decay order; the shifted camera in closed form on the device (Camera.shifted)
  "render"    render() per view (the zero-change rasterizer surface) with the same fused loss block and one-launch Adam;
              the shifted camera built the reference's way (device inverse, device->host copy, Camera constructor);
              densification statistics through this build's GaussianModel (the reference's sums without its boolean-mask
              indexing, i.e. without two host syncs per iteration)
  "unchanged" what a user who only swaps imports gets: the reference loop's CALL SEQUENCE (golden G11, recorded from
              train.py:65-202 with stand-ins: tests/golden/make_golden_trace.py) driven by the build's own
              binocular3dgs_amd/schedule.py over this build's modules: render, l1_loss / ssim / SmoothLoss (loss_utils.py),
              inverse_warp_images (graphics_utils.py), GaussianModel (training_setup -> optim.Adam, update_learning_rate,
              opacity_decay, add_densification_stats), Scene.getShiftedCamera (scene.py: the closed form SURVEY 8a-9 asks
              for).  The loop's PyTorch glue is part of the sequence: the disparity expression, the loss sum, the masked
              write into max_radii2D (three host syncs).  `unchanged_item` also reads the loss back every iteration (the
              reference's progress bar, train.py:155).
  "torch_ops" only the rasterizer swapped: render() per view, the loss block as PyTorch ops (loss.binocular_loss =
              utils/loss_utils.py + inverse_warp_images), torch.optim.Adam over six groups, opacity decay and statistics
              as the reference's PyTorch statements (what "unchanged" was before round 5)
Not part of the headline metric: BASELINE.json's metric is 6 views per iteration with given pixel gradients.
"""
from __future__ import annotations

import gc
import random
import time

import torch

LR = (0.00016, 0.0025, 0.0025 / 20.0, 0.005, 0.001, 0.05)      # model order: xyz, f_dc, f_rest, scaling, rotation, opacity
CAM_TRANS_DIST = 0.4                                            # train.py:280
OPACITY_DECAY = 0.995                                           # train.py:279


def reference_shifted(cam, trans_dist: float):
    """The COST SHAPE of the reference's Scene.getShiftedCamera (scene/__init__.py:96-115) on this build's Camera class:
    a device-side inverse of the world-to-camera matrix, the offset of the moved centre read back to the host (one sync per
    iteration) and a NEW Camera through the constructor (two host-side 4x4 inversions, three uploads, bmm, inverse)."""
    from binocular3dgs_amd.camera import Camera
    w2c = cam.world_view_transform.t().contiguous()
    moved = torch.linalg.inv(w2c) @ torch.tensor([trans_dist, 0.0, 0.0, 1.0], device=w2c.device)
    offset = (moved[:3] - cam.camera_center).cpu().numpy()
    img = None if cam.original_image is None else torch.ones_like(cam.original_image)
    return Camera(cam.R, cam.T, cam.FoVx, cam.FoVy, cam.image_width, cam.image_height, image=img, uid=cam.uid, trans=offset,
                  device=cam.device)


def _count_launches(fn, steps=3):
    """Kernel launches per call of fn (torch profiler, device-side kernel events only)."""
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
    n = 0
    for e in prof.events():
        if str(getattr(e, "device_type", "")).endswith("CUDA") and not e.name.startswith(("Memcpy", "Memset")):
            n += 1
    return n / steps


def run(dev, P: int, W: int, H: int, fov: float, surface: str, steps: int = 40, warmup: int = 8, seed: int = 0, probe=None) -> dict:
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.fused_loss import binocular_loss_fused
    from binocular3dgs_amd.gaussian_model import inverse_sigmoid
    from binocular3dgs_amd.loss import binocular_loss
    from binocular3dgs_amd.render import PipelineParams, render
    from binocular3dgs_amd.step import FusedAdam, ViewShardedStep
    assert surface in ("fused", "fused_percam", "fused_percam_one_round", "fused_graph", "render", "unchanged", "unchanged_item",
                       "torch_ops")
    model = synth.synth_model(P, seed=seed, device=dev, width=W, height=H, fovx_deg=fov)
    model.init_densification_stats()
    cams = synth.synth_cameras(W, H, fovx_deg=fov, yaws=synth.YAWS_6, device=dev)[:3]
    g = torch.Generator(device="cpu").manual_seed(seed + 11)
    gts = [torch.rand((3, H, W), generator=g).to(dev) for _ in cams]
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    rng = random.Random(seed + 5)
    seq = [(rng.randrange(3), rng.random() * CAM_TRANS_DIST * rng.choice([-1.0, 1.0])) for _ in range(warmup + 2 * steps + 16)]
    pos = [0]

    def draw():
        k, t = seq[pos[0] % len(seq)]
        pos[0] += 1
        return k, t

    if surface in ("fused", "fused_percam", "fused_percam_one_round"):
        # fused_percam: one slot pair PER INPUT CAMERA (persistent image state, i.e. an open-tile prediction per camera, for the
        # input view and for "its shifted partner" -- whose shift changes every iteration) instead of one pair shared by whichever
        # camera is drawn: what a per-camera prediction buffer behind render() could at best gain at this iteration shape
        # (VERDICT r5 item 5); fused_percam_one_round: the same slots with seg1_fraction = 0
        percam = surface != "fused"
        opt = FusedAdam(model.parameters(), LR, eps=1e-15, opacity_decay=OPACITY_DECAY, opacity_index=5, decay_first=True)
        fused = FusedRasterizer(model, W, H, num_slots=6 if percam else 2, want_means2D=False,
                                seg1_fraction=0.0 if surface == "fused_percam_one_round" else "auto")
        st = ViewShardedStep(model, [(cams[0], cams[0].shifted(0.1), 0.1)], bg, optimizer=opt, fused=fused,
                             overflow_check_every=32)
        cur = {}

        def loss_fn(i, cam, pkg, spkg, t):
            return binocular_loss_fused(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], cur["gt"],
                                        shifted_image=spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t, slot=0,
                                        unit_grad=True)

        def step():
            k, t = draw()
            v0, v1 = st.views
            v0.cam, v0.t, v1.cam, v1.t = cams[k], t, cams[k].shifted(t), t
            if percam:
                v0.slot, v1.slot = 2 * k, 2 * k + 1
            cur["gt"] = gts[k]
            st.step(loss_fn=loss_fn)
        extra = lambda: {"binning_rounds": 2 if fused.seg1_fraction > 0 else 1, "seg1_fraction": fused.seg1_fraction,   # noqa: E731
                         "repair_rate": fused.repair_rate(),
                         "two_round_disabled": str(fused.two_round_disabled) if fused.two_round_disabled else None}
    elif surface == "fused_graph":
        # the same step as ONE HIP-graph replay per iteration: the pair's cameras live in a static device block rewritten
        # in place (CameraPairSlots: one pinned upload), the shift and the learning rates are read from device memory
        # (B3gsLossIO::trans_dist_dev, B3gsAdamSegment::lr_dev), the ground-truth image is copied into a static buffer.
        # Host work per iteration: two copies and a graph launch -- the eager loop above is ~0.7 ms of host work per
        # iteration and drops to half its speed when the (shared) host is busy.
        from binocular3dgs_amd.camera import CameraPairSlots
        slots = CameraPairSlots(cams[0], 0.1)
        gt_static = gts[0].clone()
        lr_dev = torch.tensor(LR, dtype=torch.float32, device=dev)
        opt = FusedAdam(model.parameters(), LR, eps=1e-15, opacity_decay=OPACITY_DECAY, opacity_index=5, decay_first=True)
        opt.lr_device = lr_dev
        fused = FusedRasterizer(model, W, H, num_slots=2, want_means2D=False)
        st = ViewShardedStep(model, [(slots.cam, slots.shifted, 0.1)], bg, optimizer=opt, fused=fused, overflow_check_every=0)

        def loss_fn(i, cam, pkg, spkg, t):
            return binocular_loss_fused(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gt_static,
                                        shifted_image=spkg["render"], focal_x=cam.get_focal()[0], trans_dist=0.0,
                                        trans_dist_dev=slots.trans_dist_dev, slot=0, unit_grad=True)

        state = {"graph": None, "frac": None, "n": 0}

        def capture():
            for _ in range(3):                       # (settles allocations; the graph then replays fixed addresses)
                st.step(loss_fn=loss_fn)
            torch.cuda.synchronize(dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                st.step(loss_fn=loss_fn)
            torch.cuda.current_stream().wait_stream(side)
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                st.step(loss_fn=loss_fn)
            state["graph"], state["frac"] = g_, fused.seg1_fraction

        def step():
            k, t = draw()
            slots.set(cams[k], t)
            gt_static.copy_(gts[k], non_blocking=True)
            if state["graph"] is None:
                capture()
            state["graph"].replay()
            state["n"] += 1
            if state["n"] % 32 == 0:                 # the step's own protocol, outside the graph: capacity / key span /
                if fused.check_overflow():           # two-round rule (it falls back to one round when the prediction keeps
                    raise SystemExit("reference_schedule: binning capacity overflow")   # missing: re-capture then)
                if fused.seg1_fraction != state["frac"]:
                    state["graph"] = None
        extra = lambda: {"binning_rounds": 2 if fused.seg1_fraction > 0 else 1, "hip_graph": True,   # noqa: E731
                         "two_round_disabled": str(fused.two_round_disabled) if fused.two_round_disabled else None}
    elif surface == "render":
        opt = FusedAdam(model.parameters(), LR, eps=1e-15, opacity_decay=OPACITY_DECAY, opacity_index=5, decay_first=True)

        def step():
            k, t = draw()
            cam = cams[k]
            pkg = render(cam, model, pipe, bg)
            scam = reference_shifted(cam, t)
            spkg = render(scam, model, pipe, bg)
            total = binocular_loss_fused(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[k],
                                         shifted_image=spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t, slot=0,
                                         unit_grad=True)
            total.backward()
            with torch.no_grad():
                vis = pkg["visibility_filter"]
                model.update_max_radii(pkg["radii"], vis)
                model.add_densification_stats(pkg["viewspace_points"], vis)
            opt.step()
            for p in model.parameters():
                p.grad = None
        extra = lambda: {}                                                                # noqa: E731
    elif surface in ("unchanged", "unchanged_item"):
        # the reference's per-iteration call sequence (golden G11) driven by the build's own loop over the build's modules
        import types
        from binocular3dgs_amd.scene import Scene
        from binocular3dgs_amd.schedule import IterationSchedule
        model.spatial_lr_scale = 1.0
        total_iterations = 30_000
        model.training_setup(types.SimpleNamespace(
            percent_dense=0.01, position_lr_init=LR[0], position_lr_final=LR[0] / 100.0, position_lr_delay_mult=0.01,
            position_lr_max_steps=total_iterations, feature_lr=LR[1], opacity_lr=LR[5], scaling_lr=LR[3], rotation_lr=LR[4]))
        for c, gt_ in zip(cams, gts):
            c.original_image, c.gt_alpha_mask = gt_, None
        sched = IterationSchedule(model, Scene(cams, model), pipe, bg, iterations=total_iterations, shift_cam_start=20_000,
                                  binocular=True, opacity_decay_factor=OPACITY_DECAY, lambda_dssim=0.2, densify_from_iter=500,
                                  densify_until_iter=15_000, densification_interval=100,
                                  log_item=(surface == "unchanged_item"))
        clock = [25_000]                  # inside the binocular phase

        def step():
            clock[0] += 1
            if clock[0] % sched.densification_interval == 0:      # (the densification itself is not part of this loop)
                clock[0] += 1
            k, t = draw()
            sched.run_iteration(clock[0], k, t)
        extra = lambda: {"statements": "the reference loop's call sequence (golden G11) driven by "    # noqa: E731
                                       "binocular3dgs_amd/schedule.py", "loss_item": sched.log_item}
    else:      # torch_ops
        names = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")
        opt = torch.optim.Adam([{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(model.parameters(), LR, names)],
                               lr=0.0, eps=1e-15)

        def step():
            k, t = draw()
            cam = cams[k]
            pkg = render(cam, model, pipe, bg)
            scam = reference_shifted(cam, t)
            spkg = render(scam, model, pipe, bg)
            total = binocular_loss(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[k],
                                   shifted_image=spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t)[0]
            total.backward()
            with torch.no_grad():
                model._opacity.data = inverse_sigmoid(model.get_opacity * OPACITY_DECAY)   # gaussian_model.py:307-309
                vis = pkg["visibility_filter"]
                # train.py:178-179 / scene/gaussian_model.py:409-411 as the reference writes them: boolean-mask indexing, a
                # host sync per statement (this build's GaussianModel forms the same sums without it: the "render" surface)
                g = pkg["viewspace_points"].grad
                model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], pkg["radii"][vis].float())
                model.xyz_gradient_accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)
                model.denom[vis] += 1
                opt.step()
                opt.zero_grad(set_to_none=True)
        extra = lambda: {}                                                                # noqa: E731

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if probe is not None:                  # (tools/ab_interleaved.py: A/B measurements interleaved on the warmed-up step)
        return probe(step)
    best = None
    for _ in range(3):                     # (the best of three runs: the two-view loop is a few hundred microseconds of
        #                                     host work per iteration on a host shared with other jobs)
        gc.collect()
        was = gc.isenabled()
        gc.disable()
        try:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
        finally:
            if was:
                gc.enable()
        best = el if best is None else min(best, el)
    launches = _count_launches(step)
    out = {"iters_per_s": round(steps / best, 1), "ms_per_iter": round(best / steps * 1e3, 3), "views_per_iter": 2,
           "launches_per_iter": round(launches, 1), "steps": steps}
    out.update(extra())
    return out


def table(dev, fov: float, seed: int, sizes=((500_000, 800, 600), (500_000, 504, 378), (100_000, 800, 600), (100_000, 504, 378),
                                              (500_000, 400, 300)),      # (the last: DTU at --resolution 4, script/run_dtu.py:11)
          surfaces=("fused", "fused_graph", "render", "unchanged", "unchanged_item", "torch_ops"), steps: int = 40) -> dict:
    res = {"what": "train.py's own iteration shape: ONE random input view + ONE randomly shifted partner per iteration "
                   "(cameras change every step), binocular loss block, opacity decay before the optimiser step, "
                   "densification statistics, Adam; surfaces: fused = FusedRasterizer pair batch + fused loss + one-launch "
                   "Adam, eager launches; fused_graph = the same step as ONE HIP-graph replay per iteration (cameras in a "
                   "static device block, shift and learning rates read from device memory); render = render() per view (reference-built shifted camera) + fused loss + one-launch Adam; "
                   "unchanged = the reference loop's call sequence (golden G11) driven by binocular3dgs_amd/schedule.py over this build's modules (render, l1_loss / ssim / SmoothLoss / "
                   "inverse_warp_images, GaussianModel.training_setup / update_learning_rate / opacity_decay / "
                   "add_densification_stats, optimizer.step: one HIP launch each); unchanged_item = the same with the progress "
                   "bar's loss.item(); torch_ops = only the rasterizer swapped (the loss as PyTorch ops + torch.optim.Adam)"}
    for P, W, H in sizes:
        row = {}
        for s in surfaces:
            if s == "unchanged_item" and (P, W, H) != sizes[0]:
                continue                      # (the progress bar's loss.item(): measured at the first size only)
            row[s] = run(dev, P, W, H, fov, s, steps=steps, seed=seed)
            torch.cuda.empty_cache()
        res[f"P{P // 1000}k_{W}x{H}"] = row
    return res


if __name__ == "__main__":
    import json
    import sys
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sizes = ((500_000, 800, 600), (500_000, 504, 378), (100_000, 800, 600), (100_000, 504, 378))
    args = [a for a in sys.argv[1:] if not a.startswith("--surfaces=")]
    surf = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--surfaces=")]
    if args:
        sizes = tuple(tuple(int(x) for x in a.split(",")) for a in args)
    print(json.dumps(table(dev, 60.0, 0, sizes=sizes, **({"surfaces": tuple(surf[0])} if surf else {}))))
