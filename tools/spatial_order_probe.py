"""usage (GPU box): python tools/spatial_order_probe.py -- what the STORAGE ORDER of the Gaussians is worth: the headline step
(bench.Job, HIP graph) on synth(P, seed) as generated (random order) and on the same Gaussians stored in Morton order of their
positions (10 bits per axis).  Same set, same images; neighbouring tiles then read neighbouring records / scratch rows."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from binocular3dgs_amd import synth


def morton_perm(xyz):
    lo, hi = xyz.min(0).values, xyz.max(0).values
    q = ((xyz - lo) / (hi - lo).clamp(min=1e-9) * 1023.0).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True)


orig = synth.synth_gaussians
mode = {"sort": False}
def patched(*a, **k):
    p = orig(*a, **k)
    if mode["sort"]:
        perm = morton_perm(p["xyz"])
        p = {n: t[perm].contiguous() for n, t in p.items()}
    return p
synth.synth_gaussians = patched
sys.argv = [sys.argv[0], "--no-extras", "--no-pmc", "--no-cpu-baseline"] + sys.argv[1:]
args = B.parse()
B.resolve_defaults(args, 1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for rep in range(3):
    for s in (False, True):
        mode["sort"] = s
        j = B.Job(args, dev, 0, 1, False, args.gaussians, args.width, args.height, args.fov, args.views, args.scaling)
        j.prepare(5)
        el = j.timed(40)
        ms, _ = j.kernel_times(10)
        print("morton" if s else "random", round(40 / el, 1), "iters/s", {k: round(v, 4) for k, v in ms.items()}, flush=True)
        del j
        torch.cuda.empty_cache()
