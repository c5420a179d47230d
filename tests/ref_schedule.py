"""TEST INFRASTRUCTURE: the reference's training schedule (train.py:65-202), statement by statement, on a synthetic
ground-truth scene -- run once on the HIP path and once on an oracle-backed CPU path so that their PSNR can be
compared (SURVEY 8d last row: "PSNR within 0.1 dB ... vs a batch-1 run of the same schedule"; LLFF data does not
exist in this environment, so the comparison target is the build's own oracle-backed run).

Per iteration, in the reference's order:
  update_learning_rate (xyz: get_expon_lr_func, train.py:83)       oneupSHdegree every `sh_interval` (:86-87)
  one input view (:92, cycled deterministically instead of random.choice so both runs see the same sequence)
  render (:100)   binocular-shifted render + warp loss once iteration > shift_cam_start (:124-136)
  L1 + D-SSIM + disparity + alpha loss (:139-149)   backward
  opacity_decay once iteration > densify_from_iter (:171-173)   densification statistics (:178-179)
  densify_and_prune every densification_interval (:181-186, split noise shared between the runs)
  optimizer.step / zero_grad (:196-198)
"""
import math

import numpy as np
import torch

from binocular3dgs_amd import synth
from binocular3dgs_amd.gaussian_model import GaussianModel, inverse_sigmoid
from binocular3dgs_amd.loss import binocular_loss, expon_lr, psnr
from binocular3dgs_amd.render import PipelineParams
from binocular3dgs_amd.render import render as _render

LR = dict(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, feature_lr=0.0025,
          opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)   # arguments/__init__.py:75-82
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # group order of scene/gaussian_model.py:154-161


def render(cam, model, pipe, bg):
    """render() of the build.  A model on the device takes exactly what a user's call takes (the default raw-parameter node:
    pending pair forwards, adopted depth order, batched backward); a CPU model -- the oracle-backed run -- goes through the
    same statements with the oracle stand-in in the rasterizer's place, patched in for the duration of THIS call only (a
    permanent patch makes render() leave its default node for every later caller in the process: rounds 3-5 ran the HIP side
    of the lock-step tests on the statement path that way, found in round 5)."""
    if model.get_xyz.is_cuda:
        return _render(cam, model, pipe, bg)
    import binocular3dgs_amd.render as R
    from cpu_render import OracleRasterizer
    old, R.GaussianRasterizer = R.GaussianRasterizer, OracleRasterizer
    try:
        return _render(cam, model, pipe, bg)
    finally:
        R.GaussianRasterizer = old


def make_scene(W=160, H=120, P_gt=4000, seed=31, spatial_lr_scale=4.0):
    """Ground-truth Gaussians, three input cameras, ground-truth images (oracle render: the same for both runs) and a
    deterministic initial model: half of the true centres (jittered), grey, opacity 0.1 (scene/gaussian_model.py:
    124-147 initialises opacity to 0.1 and colours from the point cloud)."""
    gt = synth.synth_model(P_gt, seed=seed, device="cpu", width=W, height=H, requires_grad=False)
    with torch.no_grad():
        gt._scaling += 0.9
    cams = synth.synth_cameras(W, H, yaws=synth.YAWS_6)
    bg = torch.zeros(3)
    with torch.no_grad():
        gts = [render(c, gt, PipelineParams(), bg)["render"].clamp(0, 1) for c in cams]
    g = torch.Generator().manual_seed(seed + 1)
    idx = torch.randperm(P_gt, generator=g)[: P_gt // 2]
    n = idx.numel()
    init = dict(xyz=gt._xyz[idx] + 0.01 * torch.randn(n, 3, generator=g),
                features_dc=torch.zeros(n, 1, 3), features_rest=torch.zeros(n, 3, 3),
                scaling=gt._scaling[idx] - 0.2, rotation=torch.randn(n, 4, generator=g),
                opacity=inverse_sigmoid(0.1 * torch.ones(n, 1)))
    return dict(W=W, H=H, cams=cams, gts=gts, init=init, bg=bg, extent=spatial_lr_scale)


def _optimizer(model, extent):
    groups = [(model._xyz, LR["position_lr_init"] * extent), (model._features_dc, LR["feature_lr"]),
              (model._features_rest, LR["feature_lr"] / 20.0), (model._opacity, LR["opacity_lr"]),
              (model._scaling, LR["scaling_lr"]), (model._rotation, LR["rotation_lr"])]
    return torch.optim.Adam([{"params": [p], "lr": lr, "name": n} for (p, lr), n in zip(groups, NAMES)], lr=0.0, eps=1e-15)


def _densify_cpu(model, opt, thr, min_opacity, extent, noise):
    """CPU counterpart of binocular3dgs_amd.densify.densify_and_prune (tests/densify_ref.py is pinned by G8)."""
    from densify_ref import densify_and_prune
    attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                rotation="_rotation")
    params = {n: getattr(model, a).detach() for n, a in attr.items()}
    m, v = {}, {}
    for n, a in attr.items():
        st = opt.state.get(getattr(model, a), {})
        m[n] = st.get("exp_avg", torch.zeros_like(params[n]))
        v[n] = st.get("exp_avg_sq", torch.zeros_like(params[n]))
    new_p, new_m, new_v = densify_and_prune(params, m, v, model.xyz_gradient_accum.clone(), model.denom.clone(), thr,
                                            min_opacity, extent, None, 0.01, noise)
    for group in opt.param_groups:
        n = group["name"]
        old = group["params"][0]
        st = opt.state.pop(old, None)
        p = torch.nn.Parameter(new_p[n].contiguous(), requires_grad=True)
        setattr(model, attr[n], p)
        group["params"][0] = p
        if st:
            st["exp_avg"], st["exp_avg_sq"] = new_m[n].contiguous(), new_v[n].contiguous()
            opt.state[p] = st
    newP = model._xyz.shape[0]
    model.xyz_gradient_accum = torch.zeros((newP, 1))
    model.denom = torch.zeros((newP, 1))
    model.max_radii2D = torch.zeros((newP,))
    return newP


class Trainer:
    """The reference loop on `device` ("cuda": the HIP rasterizer + HIP densification; "cpu": the oracle + the torch
    densification), one iteration per step() so that two trainers can also be run in lockstep from a shared state."""

    def __init__(self, scene, device, iterations=300, densify_from_iter=60, densification_interval=40,
                 densify_grad_threshold=0.0002, shift_cam_start=100, sh_interval=100, cam_trans_dist=0.4,
                 opacity_decay=0.995, seed=5):
        self.__dict__.update(device=device, iterations=iterations, densify_from_iter=densify_from_iter,
                             densification_interval=densification_interval, thr=densify_grad_threshold,
                             shift_cam_start=shift_cam_start, sh_interval=sh_interval, opacity_decay=opacity_decay)
        i0 = scene["init"]
        self.model = GaussianModel.from_tensors(i0["xyz"], i0["features_dc"], i0["features_rest"], i0["scaling"],
                                                i0["rotation"], i0["opacity"], sh_degree=1, active_sh_degree=0, device=device)
        self.model.init_densification_stats()
        self.cams = synth.synth_cameras(scene["W"], scene["H"], yaws=synth.YAWS_6, device=device)[:3]
        self.gts = [g.to(device) for g in scene["gts"]]
        self.bg = scene["bg"].to(device)
        self.extent = scene["extent"]
        self.opt = _optimizer(self.model, self.extent)
        self.pipe = PipelineParams()
        rng = np.random.default_rng(seed)                     # trans_dist sequence shared by all runs
        self.shifts = (rng.random(iterations + 1) * cam_trans_dist) * rng.choice([-1.0, 1.0], iterations + 1)
        self.last_newP = None

    def mean_psnr(self):
        with torch.no_grad():
            return float(np.mean([float(psnr(render(c, self.model, self.pipe, self.bg)["render"].clamp(0, 1)[None],
                                             g[None]).mean()) for c, g in zip(self.cams, self.gts)]))

    def step(self, it):
        model, opt, extent = self.model, self.opt, self.extent
        opt.param_groups[0]["lr"] = expon_lr(it, LR["position_lr_init"] * extent, LR["position_lr_final"] * extent,
                                             lr_delay_mult=LR["position_lr_delay_mult"], max_steps=self.iterations)
        if it % self.sh_interval == 0:
            model.oneupSHdegree()
        k = (it - 1) % len(self.cams)
        cam, gt = self.cams[k], self.gts[k]
        pkg = render(cam, model, self.pipe, self.bg)
        shifted, t = None, None
        if it > self.shift_cam_start:
            t = float(self.shifts[it])
            shifted = render(cam.shifted(t), model, self.pipe, self.bg)["render"]
        total, _ = binocular_loss(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gt, shifted_image=shifted,
                                  focal_x=cam.get_focal()[0], trans_dist=t)
        total.backward()
        self.last_newP = None
        with torch.no_grad():
            if self.opacity_decay and it > self.densify_from_iter:
                model._opacity.data = inverse_sigmoid(model.get_opacity * self.opacity_decay)
            vis = pkg["visibility_filter"]
            model.update_max_radii(pkg["radii"], vis)
            model.add_densification_stats(pkg["viewspace_points"], vis)
            if it > self.densify_from_iter and it % self.densification_interval == 0:
                P = model.get_xyz.shape[0]
                noise = torch.randn(2, P, 3, generator=torch.Generator().manual_seed(1000 + it))
                if self.device == "cpu":
                    self.last_newP = _densify_cpu(model, opt, self.thr, 0.005, extent, noise)
                else:
                    from binocular3dgs_amd.densify import densify_and_prune
                    self.last_newP = densify_and_prune(model, opt, self.thr, 0.005, extent, None, noise=noise.to(self.device))
            if it < self.iterations:
                opt.step()
                opt.zero_grad(set_to_none=True)
        return float(total.detach())

    # ---- state transfer (lockstep runs): parameters, Adam moments + step, densification statistics, SH degree -----
    def get_state(self):
        m, st = self.model, []
        for g in self.opt.param_groups:
            p = g["params"][0]
            s = self.opt.state.get(p, {})
            st.append(dict(p=p.detach().cpu().clone(), m=None if not s else s["exp_avg"].cpu().clone(),
                           v=None if not s else s["exp_avg_sq"].cpu().clone(),
                           step=None if not s else float(s["step"])))
        return dict(groups=st, accum=m.xyz_gradient_accum.cpu().clone(), denom=m.denom.cpu().clone(),
                    radii=m.max_radii2D.cpu().clone(), sh=m.active_sh_degree)

    def set_state(self, state):
        m, dev = self.model, self.device
        attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                    rotation="_rotation")
        for g, s in zip(self.opt.param_groups, state["groups"]):
            old = g["params"][0]
            self.opt.state.pop(old, None)
            p = torch.nn.Parameter(s["p"].to(dev).contiguous(), requires_grad=True)
            setattr(m, attr[g["name"]], p)
            g["params"][0] = p
            if s["m"] is not None:
                self.opt.state[p] = dict(step=torch.tensor(s["step"]), exp_avg=s["m"].to(dev).contiguous(),
                                         exp_avg_sq=s["v"].to(dev).contiguous())
        m.xyz_gradient_accum, m.denom = state["accum"].to(dev), state["denom"].to(dev)
        m.max_radii2D, m.active_sh_degree = state["radii"].to(dev), state["sh"]


class SwappedTrainer(Trainer):
    """What a user who only swaps imports runs: the reference loop's call sequence (golden G11) driven by the build's own
    binocular3dgs_amd/schedule.py over this build's modules -- render, l1_loss / ssim / SmoothLoss (loss_utils),
    inverse_warp_images (graphics_utils), GaussianModel.training_setup / update_learning_rate / opacity_decay /
    add_densification_stats / densify_and_prune, gaussians.optimizer (optim.Adam), scene.getShiftedCamera: every one of them
    a HIP launch behind the reference's signature.  Same step() / get_state() surface as Trainer (its optimiser keeps
    torch's state layout), so it can lead a lockstep run."""

    def __init__(self, scene, iterations=300, **kw):
        import types
        from binocular3dgs_amd.scene import Scene
        from binocular3dgs_amd.schedule import IterationSchedule
        super().__init__(scene, "cuda", iterations=iterations, **kw)
        m = self.model
        m.spatial_lr_scale = self.extent
        m.training_setup(types.SimpleNamespace(
            percent_dense=0.01, position_lr_init=LR["position_lr_init"], position_lr_final=LR["position_lr_final"],
            position_lr_delay_mult=LR["position_lr_delay_mult"], position_lr_max_steps=iterations, feature_lr=LR["feature_lr"],
            opacity_lr=LR["opacity_lr"], scaling_lr=LR["scaling_lr"], rotation_lr=LR["rotation_lr"]))
        self.opt = m.optimizer
        for cam, gt in zip(self.cams, self.gts):
            cam.original_image, cam.gt_alpha_mask = gt, None

        def shared_split_noise(it):       # (both lock-step trainers split with the same noise)
            P = m.get_xyz.shape[0]
            m.split_noise = torch.randn(2, P, 3, generator=torch.Generator().manual_seed(1000 + it)).cuda()

        # densification statistics are kept in every iteration of these short runs (Trainer.step does the same)
        self.sched = IterationSchedule(
            m, Scene(self.cams, m, cameras_extent=self.extent), self.pipe, self.bg, iterations=self.iterations,
            shift_cam_start=self.shift_cam_start, binocular=True, opacity_decay_factor=self.opacity_decay or None,
            lambda_dssim=0.2, densify_from_iter=self.densify_from_iter, densify_until_iter=self.iterations + 1,
            densification_interval=self.densification_interval, densify_grad_threshold=self.thr, sh_interval=self.sh_interval,
            before_densify=shared_split_noise)

    def step(self, it):
        sched = self.sched
        sched.densify_until_iter = self.iterations + 1
        total = sched.run_iteration(it, (it - 1) % len(self.cams), float(self.shifts[it]))
        self.last_newP = self.model.get_xyz.shape[0] if sched.densified else None
        return float(total.detach())


class FusedTrainer:
    """The same schedule on the build's OWN step (what bench.py times, plus the real loss): FusedRasterizer (raw
    parameters, the input view and its shifted partner as one batch, shared depth sort, tight binning, in-kernel
    activations), the fused loss block (b3gs_binocular_loss), multi-view chain rule with the densification statistics
    folded in, one-launch Adam with the reference's decay order (opacity decay BEFORE the update, train.py:171-173 vs
    :196-198), HIP densification.  Same step() / get_state() / mean_psnr() surface as Trainer, so it can lead a lockstep
    run against the oracle-backed CPU trainer."""

    def __init__(self, scene, iterations=300, densify_from_iter=60, densification_interval=40,
                 densify_grad_threshold=0.0002, shift_cam_start=100, sh_interval=100, cam_trans_dist=0.4,
                 opacity_decay=0.995, seed=5, seg1_fraction="auto"):
        from binocular3dgs_amd.fused import FusedRasterizer
        from binocular3dgs_amd.step import FusedAdam, ViewShardedStep
        dev = "cuda"
        self.__dict__.update(device=dev, iterations=iterations, densify_from_iter=densify_from_iter,
                             densification_interval=densification_interval, thr=densify_grad_threshold,
                             shift_cam_start=shift_cam_start, sh_interval=sh_interval, opacity_decay=opacity_decay)
        i0 = scene["init"]
        self.model = GaussianModel.from_tensors(i0["xyz"], i0["features_dc"], i0["features_rest"], i0["scaling"],
                                                i0["rotation"], i0["opacity"], sh_degree=1, active_sh_degree=0, device=dev)
        self.model.init_densification_stats()
        W, H = scene["W"], scene["H"]
        self.cams = synth.synth_cameras(W, H, yaws=synth.YAWS_6, device=dev)[:3]
        self.gts = [g.to(dev) for g in scene["gts"]]
        self.bg = scene["bg"].to(dev)
        self.extent = scene["extent"]
        # parameter order of the model: xyz, f_dc, f_rest, scaling, rotation, opacity
        lrs = [LR["position_lr_init"] * self.extent, LR["feature_lr"], LR["feature_lr"] / 20.0, LR["scaling_lr"],
               LR["rotation_lr"], LR["opacity_lr"]]
        self.opt = FusedAdam(self.model.parameters(), lrs, eps=1e-15, opacity_decay=0.0, opacity_index=5, decay_first=True)
        self.fused = FusedRasterizer(self.model, W, H, num_slots=2, want_means2D=False, seg1_fraction=seg1_fraction)
        self.st = ViewShardedStep(self.model, [(self.cams[0], self.cams[0].shifted(0.1), 0.1)], self.bg,
                                  optimizer=self.opt, fused=self.fused, overflow_check_every=1)
        self.pipe = PipelineParams()
        rng = np.random.default_rng(seed)                     # trans_dist sequence shared by all runs
        self.shifts = (rng.random(iterations + 1) * cam_trans_dist) * rng.choice([-1.0, 1.0], iterations + 1)
        self.last_newP = None
        self._cur = {}

    def mean_psnr(self):
        with torch.no_grad():
            return float(np.mean([float(psnr(render(c, self.model, self.pipe, self.bg)["render"].clamp(0, 1)[None],
                                             g[None]).mean()) for c, g in zip(self.cams, self.gts)]))

    def _loss(self, i, cam, pkg, spkg, t):
        from binocular3dgs_amd.fused_loss import binocular_loss_fused
        use = self._cur["it"] > self.shift_cam_start
        total = binocular_loss_fused(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], self._cur["gt"],
                                     shifted_image=spkg["render"] if use else None, focal_x=cam.get_focal()[0],
                                     trans_dist=t if use else None, slot=0, unit_grad=True)
        self._cur["loss"] = total.detach()
        return total

    def step(self, it):
        model, opt, st = self.model, self.opt, self.st
        opt.lrs[0] = expon_lr(it, LR["position_lr_init"] * self.extent, LR["position_lr_final"] * self.extent,
                              lr_delay_mult=LR["position_lr_delay_mult"], max_steps=self.iterations)
        if it % self.sh_interval == 0:
            model.oneupSHdegree()
        k = (it - 1) % len(self.cams)
        t = float(self.shifts[it])
        v0, v1 = st.views
        # (before the binocular phase the shifted view is rendered but receives no gradient: the reference does not
        # render it at all, which leaves every result the same)
        v0.cam, v0.t, v1.cam, v1.t = self.cams[k], t, self.cams[k].shifted(t), t
        self._cur.update(it=it, gt=self.gts[k])
        st.compute_grads(loss_fn=self._loss)
        decay = self.opacity_decay if (self.opacity_decay and it > self.densify_from_iter) else 0.0
        self.last_newP = None
        if it > self.densify_from_iter and it % self.densification_interval == 0:
            if decay > 0.0:
                with torch.no_grad():   # the reference replaces every parameter here: optimizer.step() then updates nothing
                    model._opacity.data.copy_(inverse_sigmoid(model.get_opacity * decay))
            P = model.get_xyz.shape[0]
            noise = torch.randn(2, P, 3, generator=torch.Generator().manual_seed(1000 + it)).to(self.device)
            self.last_newP = st.densify_and_prune(self.thr, 0.005, self.extent, noise=noise)
        elif it < self.iterations:
            opt.opacity_decay = decay
            st.reduce_and_update()
        st.check_capacity()
        return float(self._cur["loss"])

    def get_state(self):
        m, opt = self.model, self.opt
        attr = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                    rotation="_rotation")
        order = ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")          # FusedAdam's flat moments
        offs, off = {}, 0
        for n in order:
            offs[n] = off
            off += getattr(m, attr[n]).numel()
        step = int(opt.step_count.item())
        groups = []
        for n in NAMES:
            p = getattr(m, attr[n])
            a, b = offs[n], offs[n] + p.numel()
            groups.append(dict(p=p.detach().cpu().clone(),
                               m=None if step == 0 else opt.exp_avg[a:b].view(p.shape).cpu().clone(),
                               v=None if step == 0 else opt.exp_avg_sq[a:b].view(p.shape).cpu().clone(),
                               step=None if step == 0 else float(step)))
        return dict(groups=groups, accum=m.xyz_gradient_accum.cpu().clone(), denom=m.denom.cpu().clone(),
                    radii=m.max_radii2D.cpu().clone(), sh=m.active_sh_degree)

    def flat_params(self):
        m = self.model
        return torch.cat([getattr(m, a).detach().reshape(-1).cpu() for a in
                          ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")])


def train_fused(scene, iterations=300, eval_every=100, **kw):
    """Free run of FusedTrainer; same history dict as train()."""
    tr = FusedTrainer(scene, iterations=iterations, **kw)
    hist = dict(psnr=[], P=[])
    for it in range(1, iterations + 1):
        tr.step(it)
        if tr.last_newP is not None:
            hist["P"].append((it, int(tr.last_newP)))
        if it % eval_every == 0 or it == iterations:
            hist["psnr"].append((it, tr.mean_psnr()))
    return hist


def train(scene, device, iterations=300, eval_every=100, **kw):
    """Free run.  Returns dict(psnr=[(iteration, mean train-view PSNR)], P=[(iteration, P after densify)])."""
    tr = Trainer(scene, device, iterations=iterations, **kw)
    hist = dict(psnr=[], P=[])
    for it in range(1, iterations + 1):
        tr.step(it)
        if tr.last_newP is not None:
            hist["P"].append((it, int(tr.last_newP)))
        if it % eval_every == 0 or it == iterations:
            hist["psnr"].append((it, tr.mean_psnr()))
    return hist
