"""cProfile of a few bench steps: where does the HOST time go?"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-timing", "--no-ohem-probe", "--no-psa-probe", "--i64-steps", "0", "--ref-steps", "0", "--fp32-steps", "0"] + sys.argv[1:]
import bench
pr = cProfile.Profile()
import torch
orig = bench.train_step
calls = {"n": 0}
def wrapped(*a, **k):
    calls["n"] += 1
    if calls["n"] == 6: pr.enable()
    r = orig(*a, **k)
    return r
bench.train_step = wrapped
bench.main()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
