"""usage (GPU box): python tools/repair_probe.py [steps] -- the headline workload with two-round binning: per step and
view, the image-header words (N1, V, N2, tiles left open and not predicted) and the number of tiles predicted open --
shows how quickly the open-tile prediction settles and how often the repair kernel has work."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sys.argv = sys.argv[:1]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    job = bench.Job(args, dev, 0, 1, False, args.gaussians, args.width, args.height, args.fov, 6, "weak",
                    seg1_fraction=float(os.environ.get("PROBE_FRAC", "0.125")))
    fr = job.fused
    L = __import__("binocular3dgs_amd._lib", fromlist=["lib"]).lib()
    for it in range(steps):
        job.eager_step()
        torch.cuda.synchronize()
        rows = []
        for s in fr.slots[:job.local_views]:
            hdr = s.img[:64 * 4].view(torch.int32)[:12].tolist()
            rows.append((hdr[0], hdr[2], hdr[3], hdr[8], hdr[9]))
        print(it, rows, flush=True)

main()
