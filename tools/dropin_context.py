"""usage (GPU box): python tools/dropin_context.py [fused] -- the zero-change surface measured the way bench.py's extras do
(Job(path="dropin"), prepare(w), timed_best(k)), alone or behind a fused Job in the same process: does the context of the
other extras change the number?"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from binocular3dgs_amd import rasterizer as R

sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a.startswith("--")]
first = os.environ.get("CTX", "")
args = B.parse()
B.resolve_defaults(args, 1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
P, W, H = args.gaussians, args.width, args.height
if "fused" in first:
    j = B.Job(args, dev, 0, 1, False, P, W, H, args.fov, 6, "weak")
    j.prepare(3)
    print("fused", round(10 / j.timed_best(10), 1))
    if "ktimes" in first:
        j.kernel_times(5)
    del j
    torch.cuda.empty_cache()
dargs = argparse.Namespace(**vars(args))
dargs.optimizer = "b3gs"
j = B.Job(dargs, dev, 0, 1, False, P, W, H, args.fov, 6, "weak", path="dropin", graph=False)
j.prepare(int(os.environ.get("WARM", "2")))
for rep in range(3):
    print("dropin lazy ", round(10 / j.timed_best(10), 1))
R._flush_pending()
R._LAZY_FWD = False
for rep in range(2):
    print("dropin eager", round(10 / j.timed_best(10), 1))
