"""TEST INFRASTRUCTURE for golden G11 (tests/golden/train_trace.json): recording stand-ins for everything the per-iteration
loop of a Binocular3DGS trainer CALLS, plus a TorchFunctionMode that follows tensors from one call's outputs to the next
call's inputs.  Nothing here computes a render or a loss: every stand-in returns fresh leaf tensors of fixed value, which is
what makes the recorded sequence a function of the LOOP alone:

  * which callables run, in which order, with which argument shapes / dtypes / scalar constants;
  * which earlier outputs each tensor argument was derived from ("from": labels like "render#0.rendered_depth");
  * the mean of every derived tensor argument (pins glue arithmetic such as the disparity expression);
  * after backward(): d(total loss) / d(each loss stand-in's output) -- the weights of the loss sum;
  * where the loop synchronises with the device (`.item()`) and where it writes model state by mask (`setitem`).

The same stand-ins are (a) patched into the reference's train module by tests/golden/make_golden_trace.py -- in the build
container only, the reference's Python never travels -- and (b) handed to this build's own driver
(binocular3dgs_amd/schedule.py) by tests/test_schedule_trace.py, which asserts that both produce the same events.
"""
from __future__ import annotations

import contextlib
import types

import torch

RENDER_VALUES = {"render": 0.25, "rendered_depth": 2.0, "rendered_alpha": 0.5}
LOSS_VALUES = {"l1_loss": 2.0, "l1_loss_masked": 7.0, "ssim": 3.0, "smooth_loss": 5.0}
WARP_VALUE = 0.125
FOCAL = (55.0, 56.5)
P_POINTS = 23
EXTENT = 3.5


class Recorder:
    def __init__(self, H=12, W=16):
        self.H, self.W = H, W
        self.iterations = []          # list of event lists
        self.setup = []               # what runs before the first iteration (constructors)
        self.events = self.setup
        self.tags = {}                # id(tensor) -> (tensor kept alive, frozenset of labels)
        self.counts = {}
        self.loss_leaves = []         # [(label, leaf)] of the iteration in flight
        self.muted = 0

    # ---- bookkeeping ------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def quiet(self):
        self.muted += 1
        try:
            yield
        finally:
            self.muted -= 1

    def begin_iteration(self, number):
        self.events = []
        self.iterations.append({"iteration": int(number), "events": self.events})
        self.counts = {}
        self.loss_leaves = []

    def tag(self, t, labels):
        if isinstance(t, torch.Tensor):
            old = self.tags.get(id(t))
            have = old[1] if old is not None and old[0] is t else frozenset()
            self.tags[id(t)] = (t, have | frozenset(labels))
        elif isinstance(t, (tuple, list)):
            for x in t:
                self.tag(x, labels)
        elif isinstance(t, dict):
            for x in t.values():
                self.tag(x, labels)

    def labels(self, t):
        ent = self.tags.get(id(t))
        return ent[1] if ent is not None and ent[0] is t else frozenset()

    def sources(self, *objs):
        out = set()
        for o in objs:
            if isinstance(o, torch.Tensor):
                out |= self.labels(o)
            elif isinstance(o, (tuple, list)):
                out |= self.sources(*o)
            elif isinstance(o, dict):
                out |= self.sources(*o.values())
        return out

    def occurrence(self, name):
        k = self.counts.get(name, 0)
        self.counts[name] = k + 1
        return f"{name}#{k}"

    def describe(self, v):
        if isinstance(v, torch.Tensor):
            d = {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "from": sorted(self.labels(v))}
            if v.numel() and (v.is_floating_point() or v.dtype in (torch.int32, torch.int64, torch.bool)):
                with self.quiet(), torch.no_grad():
                    d["mean"] = round(float(v.detach().double().mean()), 6)
            d["requires_grad"] = bool(v.requires_grad)
            return d
        if isinstance(v, CameraStandIn):
            return {"camera": v.uid, "shift": v.shift}
        if isinstance(v, ModelStandIn):
            return "model"
        if isinstance(v, (bool, int, float, str)) or v is None:
            return round(v, 9) if isinstance(v, float) else v
        if isinstance(v, types.SimpleNamespace) or hasattr(v, "convert_SHs_python") or hasattr(v, "__dict__"):
            return type(v).__name__ if not isinstance(v, types.SimpleNamespace) else "namespace"
        return repr(type(v))

    def event(self, call, **fields):
        if self.events is None or self.muted:
            return None
        ev = {"call": call}
        ev.update(fields)
        self.events.append(ev)
        return ev

    def call(self, name, args=(), kwargs=None, names=()):
        """Record a call; returns its occurrence label ("render#1")."""
        occ = self.occurrence(name)
        desc = {}
        for i, a in enumerate(args):
            desc[names[i] if i < len(names) else f"arg{i}"] = self.describe(a)
        for k, a in (kwargs or {}).items():
            desc[k] = self.describe(a)
        self.event(occ, args=desc)
        return occ

    def leaf(self, shape, value, label, dtype=torch.float32, requires_grad=True):
        with self.quiet():
            t = torch.full(tuple(shape), value, dtype=dtype)
            if requires_grad:
                t.requires_grad_(True)
        self.tag(t, [label])
        return t

    # ---- hooks the flow mode calls ------------------------------------------------------------------------------------
    def on_backward(self, root):
        weights = {}
        with self.quiet():
            for label, leaf in self.loss_leaves:
                weights[label] = None if leaf.grad is None else round(float(leaf.grad), 6)
        self.event("backward", root_from=sorted(self.labels(root)), weights=weights)


class Flow(torch.overrides.TorchFunctionMode):
    """Follows labels through torch functions; records backward(), .item() and masked writes into labelled model state."""

    def __init__(self, rec: Recorder):
        super().__init__()
        self.rec = rec

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = kwargs or {}
        rec = self.rec
        out = func(*args, **kwargs)
        if rec.muted:
            return out
        if func is torch.Tensor.backward:
            rec.on_backward(args[0])
            return out
        src = rec.sources(args, kwargs)
        if func is torch.Tensor.item:
            rec.event("item", of=sorted(src))
        elif func is torch.Tensor.__setitem__:
            tgt = sorted(rec.labels(args[0]))
            if any(t.startswith("model.") for t in tgt):
                rec.event("setitem", target=tgt, index_from=sorted(rec.sources(args[1])),
                          value_from=sorted(rec.sources(args[2])))
        elif src:
            rec.tag(out, src)
        return out


# ---- stand-ins ------------------------------------------------------------------------------------------------------
class CameraStandIn:
    def __init__(self, rec, uid, shift=None):
        self.uid, self.shift = uid, shift
        self.image_height, self.image_width = rec.H, rec.W
        with rec.quiet():
            self.original_image = torch.full((3, rec.H, rec.W), 0.4 + 0.1 * (uid if isinstance(uid, int) else 0))
        rec.tag(self.original_image, [f"cam{uid}.original_image"])
        self.gt_alpha_mask = None

    def get_focal(self):
        return FOCAL


class OptimizerStandIn:
    def __init__(self, rec):
        self.rec = rec

    def step(self, *a, **k):
        self.rec.call("optimizer.step", a, k)

    def zero_grad(self, *a, **k):
        self.rec.call("optimizer.zero_grad", a, k)


class ModelStandIn:
    """Stands where the trainer's GaussianModel stands: every per-iteration method is recorded."""

    def __init__(self, rec, sh_degree=1):
        self.rec = rec
        self.max_sh_degree = sh_degree
        self.optimizer = None
        with rec.quiet():
            self.max_radii2D = torch.zeros(P_POINTS)
            self._xyz = torch.zeros(P_POINTS, 3)
        rec.tag(self.max_radii2D, ["model.max_radii2D"])

    @property
    def get_xyz(self):
        return self._xyz

    def training_setup(self, training_args):
        self.optimizer = OptimizerStandIn(self.rec)

    def update_learning_rate(self, iteration):
        self.rec.begin_iteration(iteration)          # (the first call of every iteration: train.py:83)
        self.rec.call("update_learning_rate", (iteration,), names=("iteration",))

    def oneupSHdegree(self):
        self.rec.call("oneupSHdegree")

    def opacity_decay(self, *a, **k):
        self.rec.call("opacity_decay", a, k, names=("factor",))

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.rec.call("add_densification_stats", (viewspace_point_tensor, update_filter), names=("viewspace_points", "filter"))

    def densify_and_prune(self, *a, **k):
        self.rec.call("densify_and_prune", a, k, names=("max_grad", "min_opacity", "extent", "max_screen_size"))


class SceneStandIn:
    def __init__(self, rec, model, n_cameras=3):
        self.rec, self.gaussians = rec, model
        self.cameras_extent = EXTENT
        self.model_path = ""
        self.cameras = [CameraStandIn(rec, uid) for uid in range(n_cameras)]

    def getTrainCameras(self, scale=1.0):
        return self.cameras

    def getTestCameras(self, scale=1.0):
        return []

    def getShiftedCamera(self, camera, trans_dist=0.1):
        self.rec.call("getShiftedCamera", (camera, trans_dist), names=("camera", "trans_dist"))
        return CameraStandIn(self.rec, camera.uid, shift=round(float(trans_dist), 9))

    def save(self, iteration):
        pass


def make_callables(rec: Recorder):
    """-> namespace(render, l1_loss, ssim, SmoothLoss, inverse_warp_images): the five free callables of the loop."""

    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
        occ = rec.call("render", (viewpoint_camera, pc, pipe, bg_color), names=("camera", "model", "pipe", "bg"))
        H, W = viewpoint_camera.image_height, viewpoint_camera.image_width
        pkg = {"render": rec.leaf((3, H, W), RENDER_VALUES["render"], f"{occ}.render"),
               "rendered_depth": rec.leaf((1, H, W), RENDER_VALUES["rendered_depth"], f"{occ}.rendered_depth"),
               "rendered_alpha": rec.leaf((1, H, W), RENDER_VALUES["rendered_alpha"], f"{occ}.rendered_alpha"),
               "viewspace_points": rec.leaf((P_POINTS, 3), 0.0, f"{occ}.viewspace_points")}
        with rec.quiet():
            radii = (torch.arange(P_POINTS) % 7).to(torch.int32)
            vis = radii > 0
        rec.tag(radii, [f"{occ}.radii"])
        rec.tag(vis, [f"{occ}.visibility_filter"])
        pkg["radii"], pkg["visibility_filter"] = radii, vis
        return pkg

    def _loss(name, value, args, kwargs, names):
        occ = rec.call(name, args, kwargs, names=names)
        out = rec.leaf((), value, f"{occ}.value")
        rec.loss_leaves.append((occ, out))
        return out

    def l1_loss(network_output, gt, mask=None):
        if mask is None:
            return _loss("l1_loss", LOSS_VALUES["l1_loss"], (network_output, gt), None, ("network_output", "gt"))
        return _loss("l1_loss", LOSS_VALUES["l1_loss_masked"], (network_output, gt), {"mask": mask}, ("network_output", "gt"))

    def ssim(img1, img2, window_size=11, size_average=True):
        return _loss("ssim", LOSS_VALUES["ssim"], (img1, img2), None, ("img1", "img2"))

    class SmoothLoss:
        def __init__(self):
            rec.call("SmoothLoss")

        def forward(self, disparity, image):
            return _loss("smooth_loss", LOSS_VALUES["smooth_loss"], (), {"disparity": disparity, "image": image}, ())

        __call__ = forward

    def inverse_warp_images(image, disparity, row_indices=None, column_indices=None):
        occ = rec.call("inverse_warp_images", (image, disparity, row_indices, column_indices),
                       names=("image", "disparity", "row_indices", "column_indices"))
        return rec.leaf(tuple(image.shape), WARP_VALUE, f"{occ}.out")

    return types.SimpleNamespace(render=render, l1_loss=l1_loss, ssim=ssim, SmoothLoss=SmoothLoss,
                                 inverse_warp_images=inverse_warp_images)


def strip_optional(iterations, keep_item=False):
    """The events a driver must reproduce.  An `.item()` of nothing recorded (`of == []`) is the reference drawing its random
    baseline on the host (train.py:125-126): the driver is GIVEN the draw.  The `.item()` of the loss (the progress bar,
    train.py:155, a device read-back per iteration) is reproduced only on request."""
    out = []
    for it in iterations:
        evs = [e for e in it["events"] if e["call"] != "item" or (keep_item and e["of"])]
        out.append({"iteration": it["iteration"], "events": evs})
    return out
