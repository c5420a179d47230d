import cProfile, pstats, sys, torch
sys.path.insert(0, "/root/repo")
import bench_ref_schedule as B
dev = torch.device("cuda", 0)
# monkeypatch run's timed loop: reuse run() with small steps under cProfile
pr = cProfile.Profile()
pr.enable()
r = B.run(dev, 500000, 800, 600, 60.0, "render", steps=60, warmup=8)
pr.disable()
print(r)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
