"""usage (GPU box): python tools/dropin_host_profile.py -- cProfile of the zero-change surface's iteration on a scene small
enough for the device to be idle (3000 Gaussians, 128x96): where the HOST time of render() / backward() / Adam goes."""
import argparse, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

sys.argv = [sys.argv[0], "--gaussians", "3000", "--width", "128", "--height", "96"]
args = B.parse()
B.resolve_defaults(args, 1)
args.optimizer = "b3gs"
dev = torch.device("cuda", 0)
j = B.Job(args, dev, 0, 1, False, 3000, 128, 96, args.fov, 6, "weak", path="dropin", graph=False)
j.prepare(5)
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(200):
    j.eager_step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
