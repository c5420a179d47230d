#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (BASELINE.json):

    train iters/s (fwd+bwd) + Mpix/s, 1M Gaussians @ 800x600, 1/2/4/8 GPU

One *iter* = 6 views = 3 input views + their 3 binocular-shifted partners, each rasterised forward
and backward through the drop-in `render()` (activations and their autograd included), the
per-Gaussian gradients of all views accumulated in one flat slab, all-reduced over RCCL when
N > 1 (weak scaling: every rank renders its own 6 views, distinct yaw offsets), and one Adam step.
Inputs are the seeded synthetic scene of BASELINE.md section 3, resident in HBM before the timed
region; upstream pixel gradients are the seeded N(0,1)/(3HW), /HW, /HW tensors of SURVEY 8(d)
(primary views: colour+depth+alpha; shifted views: colour only, as the loss block of
train.py:123-149 produces).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel (blend backward): algorithmic bytes / HIP-event duration vs 8 TB/s
  roofline_view whole view fwd+bwd: SURVEY 8(d) byte model (R+W) / per-view kernel time
  cpu_baseline  oracle/tile_ref.c (kind "port") on the host cores, one full-size view fwd+bwd
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--optimizer", choices=("b3gs", "torch"), default="b3gs",
                    help="b3gs: one-launch fused Adam (b3gs_adam_step); torch: torch.optim.Adam(fused=True), 12 launches")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--path", choices=["fused", "dropin"], default="fused",
                    help="fused: b3gs_forward_raw/backward_raw (activations in-kernel, persistent scratch, no host sync); "
                         "dropin: the reference-shaped render() -> _C.rasterize_gaussians surface")
    ap.add_argument("--schedule", choices=("batched", "streams", "serial"), default="batched",
                    help="batched: every stage one launch for all 6 views, pairs share a depth sort; "
                         "streams: one stream per view; serial: one stream, per-view launches")
    ap.add_argument("--pipeline-ranges", type=int, default=0,
                    help="data parallel: cut the chain-rule / all-reduce / Adam tail into this many Gaussian ranges so the "
                         "all-reduce of one range overlaps the neighbours' compute (0: one all-reduce of the whole slab). "
                         "Off by default: on ONE GPU the extra launches cost 0.2 ms per iteration (2.39 -> 2.59 ms), "
                         "about what the overlap can win back at 8 GPUs")
    ap.add_argument("--loss", choices=("synthetic", "fused", "torch"), default="synthetic",
                    help="synthetic: fixed pixel gradients (the metric's definition); fused / torch: the loss block of "
                         "train.py:123-148 through b3gs_binocular_loss / through PyTorch ops")
    ap.add_argument("--viewspace-grads", action="store_true",
                    help="also write every view's [P,3] screen-space mean gradients (viewspace_points.grad)")
    ap.add_argument("--serial-views", action="store_true",
                    help="render the views one after the other on one stream (un-overlapped kernel times, for profiles)")
    ap.add_argument("--dp-path", action="store_true",
                    help="single-GPU check of the N>1 code path: 1-rank RCCL group, graph + eager all-reduce")
    ap.add_argument("--graph", type=int, default=1, help="capture one whole iteration in a HIP graph (fused path only)")
    return ap.parse_args()


def byte_model(P, V, N, HW, Tn, K=4, tiles_bits=None):
    """SURVEY.md 8(d) algorithmic bytes per view, fwd+bwd, K SH coeffs, p radix byte-passes of the
    reference's 64-bit sort (the model is the REFERENCE algorithm's traffic: what a perfect
    implementation of that algorithm must move; our two-level sort moves less)."""
    p = math.ceil((32 + max(1, math.ceil(math.log2(max(Tn, 2))))) / 8)
    R = 12 * P + (32 + 12 * K) * V + 4 * P + 20 * V + (12 * p + 8) * N + 8 * N + 44 * N + 8 * Tn + 44 * N + 28 * HW + \
        (44 + 12 * K + 76) * V
    Wt = 8 * P + 68 * V + 4 * P + 12 * N + 12 * p * N + 8 * Tn + 28 * HW + 48 * P + 48 * V + (40 + 12 * K) * P
    return R, Wt


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists for the rasterizer)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = world > 1 or args.dp_path
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from binocular3dgs_amd import _lib, synth
    from binocular3dgs_amd.render import PipelineParams
    from binocular3dgs_amd.step import ViewShardedStep

    P, W, H = args.gaussians, args.width, args.height
    model = synth.synth_model(P, seed=args.seed, device=dev, width=W, height=H)
    # weak scaling: each rank owns 3 distinct pairs (yaw offsets 0, 1.5, 3.0 ... degrees apart)
    pairs = synth.synth_view_set(W, H, device=dev, yaw_offset=1.5 * rank)
    bg = torch.zeros(3, device=dev)
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=rank, device=dev)
    gc2 = synth.synth_pixel_grads(W, H, seed=100 + rank, device=dev)[0]
    opt = None
    if not args.no_optimizer:
        # learning rates of arguments/__init__.py:75-82, eps of scene/gaussian_model.py:163
        lrs = [1.6e-4, 2.5e-3, 2.5e-3 / 20, 5e-3, 1e-3, 0.05]
        if args.optimizer == "b3gs":
            from binocular3dgs_amd.step import FusedAdam
            opt = FusedAdam(model.parameters(), lrs, eps=1e-15)
        else:
            opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(model.parameters(), lrs)], lr=0.0,
                                   eps=1e-15, fused=True)
    fused = None
    if args.path == "fused":
        from binocular3dgs_amd.fused import FusedRasterizer
        # the densification statistics (train.py:178-179: the only consumer of the screen-space gradients) are
        # updated inside the per-Gaussian backward pass, so the per-view [P,3] gradient tensors are not
        # materialised unless asked for
        model.init_densification_stats()
        fused = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=bool(args.viewspace_grads),
                                schedule="serial" if args.serial_views else args.schedule)
    # data parallel: the tail of the iteration (chain rule -> all-reduce -> Adam) is pipelined over Gaussian ranges
    pipe_ranges = args.pipeline_ranges if (dp or args.dp_path) and fused is not None and args.optimizer == "b3gs" else 0
    stepper = ViewShardedStep(model, pairs, bg, PipelineParams(), optimizer=opt, fused=fused, pipeline_ranges=pipe_ranges)
    stepper.slab.force_collective = bool(args.dp_path)

    def grad_fn(i, pkg, spkg):
        out = [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga)]
        if spkg is not None:
            out.append((spkg["render"], gc2))
        return out

    step_kw = dict(pair_grad_fn=grad_fn)
    if args.loss != "synthetic":
        # the real loss block of train.py:123-148 on random ground-truth images instead of fixed pixel gradients
        # (information only: BASELINE.json's metric is the rasterizer fwd+bwd with given pixel gradients)
        from binocular3dgs_amd.fused_loss import binocular_loss_fused
        from binocular3dgs_amd.loss import binocular_loss
        gts = [torch.rand(3, H, W, device=dev) for _ in pairs]
        bgm = [(g_.max(0, keepdim=True).values < 0.1).float() for g_ in gts]

        def loss_fn(i, cam, pkg, spkg, t):
            kw = dict(shifted_image=None if spkg is None else spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t,
                      bg_mask=bgm[i])
            if args.loss == "fused":
                return binocular_loss_fused(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[i], slot=i,
                                            unit_grad=True, **kw)
            return binocular_loss(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[i], **kw)[0]
        step_kw = dict(loss_fn=loss_fn)
        if args.loss == "fused":
            from binocular3dgs_amd.fused_loss import binocular_loss_fused_batch

            def batch_loss_fn(items):
                return binocular_loss_fused_batch(
                    [dict(image=pkg["render"], depth=pkg["rendered_depth"], alpha=pkg["rendered_alpha"], gt_image=gts[i],
                          shifted_image=None if spkg is None else spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t,
                          bg_mask=bgm[i]) for i, cam, pkg, spkg, t in items], unit_grad=True)
            step_kw = dict(batch_loss_fn=batch_loss_fn)

    def barrier():
        torch.cuda.synchronize(dev)
        if dp:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        stepper.step(**step_kw)
    if fused is not None:
        while fused.overflowed():          # persistent binning capacity too small: grow once, outside the timed region
            fused.grow()
            stepper.step(**step_kw)
    run_step = lambda: stepper.step(**step_kw)  # noqa: E731
    use_graph = bool(args.graph) and fused is not None
    if use_graph and dp:
        # data parallel: the rendering part of the iteration is one hipGraph; the RCCL all-reduce of
        # the gradient slab and the (single-kernel) fused Adam are issued eagerly after each replay
        try:
            sg = torch.cuda.Stream()
            sg.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sg):
                stepper.compute_grads(**step_kw)
            torch.cuda.current_stream().wait_stream(sg)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                stepper.compute_grads(**step_kw)

            def run_step():
                graph.replay()
                stepper.reduce_and_update()
        except Exception as exc:  # capture unsupported in this environment: stay eager
            if rank == 0:
                print(f"[bench] graph capture failed ({exc!r}); running eager", file=sys.stderr)
            use_graph = False
            run_step = lambda: stepper.step(**step_kw)  # noqa: E731
    elif use_graph:
        # the fused path never allocates, never syncs and keeps N on the device: the whole iteration
        # (6 views fwd+bwd, slab zero, Adam) is one hipGraph launch
        if opt is not None and args.optimizer == "torch":
            for g_ in opt.param_groups:
                g_["capturable"] = True
            for st_ in opt.state.values():
                if "step" in st_ and not st_["step"].is_cuda:
                    st_["step"] = st_["step"].to(dev)
        sg = torch.cuda.Stream()
        sg.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sg):
            stepper.step(**step_kw)
        torch.cuda.current_stream().wait_stream(sg)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            stepper.step(**step_kw)
        run_step = graph.replay

    # stage timing: HIP events recorded by the library on the launch stream, no sync inside.
    # When the iteration is replayed from a HIP graph the library is not re-entered, so the events
    # are taken from eager iterations of the same workload run right after the timed region.
    L = _lib.lib()
    times = _lib.B3gsKernelTimes()
    barrier()
    if not use_graph:
        L.b3gs_set_timing(C.byref(times))
    views = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
        views += 2 * len(pairs)
    barrier()
    t1 = time.perf_counter()
    timed_views = args.steps * 2 * len(pairs)
    if use_graph or (fused is not None and fused.schedule == "streams"):
        # per-kernel durations need eager launches on one stream: a few more iterations after the timed
        # region, same workload and kernels ("batched" keeps its batched launches; "streams" cannot be
        # timed per kernel while overlapped and is issued view by view here)
        L.b3gs_timing_collect()            # resolve (and discard) stages parked during the timed region
        L.b3gs_set_timing(None)
        times = _lib.B3gsKernelTimes()
        L.b3gs_set_timing(C.byref(times))
        was = fused.schedule
        if was == "streams":
            fused.schedule, fused.concurrent = "serial", False
        for _ in range(min(args.steps, 5)):
            stepper.step(**step_kw)
        barrier()
        fused.schedule, fused.concurrent = was, was == "streams"
        timed_views = min(args.steps, 5) * 2 * len(pairs)
    L.b3gs_timing_collect()
    L.b3gs_set_timing(None)
    elapsed = t1 - t0
    if fused is not None and fused.overflowed():
        raise SystemExit("binning capacity overflow inside the timed region: result invalid")
    if dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # workload statistics of this rank's primary view 0 (V, N make the byte model concrete)
    with torch.no_grad():
        from binocular3dgs_amd.render import render
        pkg = render(pairs[0][0], model, PipelineParams(), bg)
        V = int((pkg["radii"] > 0).sum().item())
    n_views = max(timed_views, 1)
    # (pixel, Gaussian) pairs the sequential algorithm visits: sum over pixels of the position of their last
    # contributor (n_contrib) -- the work unit of the blend kernels (they are not HBM bound)
    pairs_per_view = None
    if fused is not None:
        from binocular3dgs_amd.debug import state_views
        s0 = fused.slots[0]
        n0 = fused.num_rendered()[0]
        pairs_per_view = int(state_views(P, W, H, n0, s0.geom, s0.binning, s0.img)["n_contrib"].to(torch.int64).sum().item())
    # N: read back from one more forward through the C ABI surface
    from binocular3dgs_amd import _C
    with torch.no_grad():
        cam = pairs[0][0]
        N = _C.rasterize_gaussians(bg, model.get_xyz, torch.empty(0, device=dev), model.get_opacity,
                                   model.get_scaling, model.get_rotation, 1.0, torch.empty(0, device=dev),
                                   cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx / 2),
                                   math.tan(cam.FoVy / 2), H, W, model.get_features, model.active_sh_degree,
                                   cam.camera_center, False, False)[0]

    iters = args.steps * world
    value = iters / elapsed
    views_per_iter = 2 * len(pairs)
    mpix = views_per_iter * W * H * iters / elapsed / 1e6

    if rank == 0:
        HW = W * H
        Tn = ((W + 15) // 16) * ((H + 15) // 16)
        ms = dict(preprocess=times.preprocess_ms / n_views, sort=times.sort_ms / n_views,
                  render_fwd=times.render_fwd_ms / n_views, render_bwd=times.render_bwd_ms / n_views,
                  preprocess_bwd=times.preprocess_bwd_ms / n_views)
        # dominant kernel = the largest stage that is a single kernel launch
        single = {"render_fwd": ms["render_fwd"], "render_bwd": ms["render_bwd"]}
        dom = max(single, key=single.get)
        # algorithmic bytes per launch (DESIGN.md "Kernels"): the blend kernels read one 44-byte record
        # + 4-byte index per tile instance; fwd writes 28 B/pixel, bwd reads 28 B/pixel and updates
        # 10 fp32 accumulators per visible Gaussian (read+write = 80 B)
        dom_bytes = (48 * N + 28 * HW + 8 * Tn) if dom == "render_fwd" else (48 * N + 28 * HW + 8 * Tn + 80 * V)
        # views per launch of that kernel: the blend backward always takes all views of the iteration, the
        # blend forward does unless every view runs on its own stream
        vpl = views_per_iter if (fused is not None and (dom == "render_bwd" or fused.schedule != "streams")) else 1
        vpl = min(vpl, 8)
        dom_s = single[dom] / 1e3
        achieved = dom_bytes / dom_s / 1e9 if dom_s > 0 else 0.0
        R, Wt = byte_model(P, V, N, HW, Tn)
        view_ms = sum(ms.values())
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                # per launch of the default schedule (all views of the iteration in one launch)
                traffic = json.load(open(tpath)).get(dom.replace("_kernel", "")) if (vpl == views_per_iter and (P, W, H) == (1_000_000, 800, 600)) else None
            except Exception:
                traffic = None
        out = {
            "metric": "train iters/s (fwd+bwd), 1M Gaussians @ 800x600, 6 views/iter",
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mpix_per_s": round(mpix, 1),
            "config": {"workload": f"synth(P={P}, seed={args.seed}) {W}x{H}, 3 input + 3 binocular-shifted views "
                                   f"per rank per iter, fwd+bwd+grad all-reduce+Adam", "gaussians": P, "width": W,
                       "height": H, "views_per_rank": views_per_iter, "global_views": views_per_iter * world,
                       "sh_degree": 1, "K": 4, "visible_V": V, "instances_N": N,
                       "instances_N_binned": (fused.num_rendered()[0] if fused is not None else N),
                       "loss": args.loss, "optimizer_in_step": opt is not None, "densify_stats_in_step": fused is not None,
                       "optimizer": None if opt is None else args.optimizer, "path": args.path, "hip_graph": bool(use_graph),
                       "schedule": None if fused is None else fused.schedule,
                       "dp_tail_ranges": pipe_ranges, "parallelism": f"dp{world} (views sharded, params replicated)"},
            "stage_ms_per_view": {k: round(v, 4) for k, v in ms.items()},
            "roofline": {"kernel": dom + "_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "bytes_per_launch": dom_bytes * vpl,
                         "avg_launch_ms": round(single[dom] * vpl, 4), "views_per_launch": vpl,
                         "note": "blend kernels are VALU-bound (SURVEY 8d caveat): see pixgauss_pairs_per_s",
                         "pixgauss_pairs_per_view": pairs_per_view,
                         "pixgauss_pairs_per_s": (None if not pairs_per_view or ms["render_bwd"] <= 0
                                                  else round(pairs_per_view / (ms["render_bwd"] / 1e3), 1))},
            "roofline_view": {"bound": "hbm", "bytes_per_view": R + Wt, "read_bytes_per_view": R,
                              "kernel_ms_per_view": round(view_ms, 4),
                              "achieved": round((R + Wt) / (view_ms / 1e3) / 1e9, 1) if view_ms > 0 else 0.0,
                              "read_frac_of_peak": round(R / (view_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if view_ms > 0 else 0.0,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s"},
        }
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N=1 only: the other ranks would wait ~7 s at the barrier
            out["cpu_baseline"] = cpu_baseline(P, W, H, args.seed)
        result_line = json.dumps(out)
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so the JSON is the LAST line
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(result_line, flush=True)


def cpu_baseline(P, W, H, seed):
    """The oracle (C port, OpenMP) on the host cores: ONE full-size view, forward + backward, same
    synthetic inputs (1/6 of an iteration).  Reported baseline only -- never on the product path."""
    from binocular3dgs_amd import synth
    from oracle import tile_ref
    cores = os.cpu_count() or 1
    model = synth.synth_model(P, seed=seed, device="cpu", width=W, height=H, requires_grad=False)
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=0)
    gc2 = synth.synth_pixel_grads(W, H, seed=100)[0]
    with torch.no_grad():
        base = dict(means3D=model.get_xyz.numpy(), opacities=model.get_opacity.numpy(),
                    scales=model.get_scaling.numpy(), rotations=model.get_rotation.numpy(),
                    shs=model.get_features.numpy(), bg=[0.0, 0.0, 0.0], W=W, H=H, sh_degree=1, threads=cores)
    fwd_s = bwd_s = 0.0
    nviews = 0
    for cam, scam, _t in synth.synth_view_set(W, H):
        for c, grads in ((cam, (gc.numpy(), gd.numpy(), ga.numpy())), (scam, (gc2.numpy(), None, None))):
            kw = dict(base, viewmatrix=c.world_view_transform.numpy(), projmatrix=c.full_proj_transform.numpy(),
                      campos=c.camera_center.numpy(), tanfovx=math.tan(c.FoVx / 2), tanfovy=math.tan(c.FoVy / 2))
            t0 = time.perf_counter()
            st = tile_ref.forward(**kw)
            t1 = time.perf_counter()
            tile_ref.backward(st, *grads)
            t2 = time.perf_counter()
            fwd_s += t1 - t0
            bwd_s += t2 - t1
            nviews += 1
    it_s = fwd_s + bwd_s
    return {"value": round(1.0 / it_s, 5), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"1 full iteration = {nviews} views fwd+bwd at full size P={P} {W}x{H} (no all-reduce, no Adam): "
                      f"fwd {fwd_s:.2f}s bwd {bwd_s:.2f}s; oracle/tile_ref.c, OpenMP {cores} threads",
            "ms_per_view": round(it_s / nviews * 1e3, 1)}


if __name__ == "__main__":
    main()
