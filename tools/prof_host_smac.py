import cProfile, pstats, sys, io, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py", "--config", "smac3s5z", "--cpu-cols", "0", "--instr-steps", "0", "--steps", "20", "--warmup", "3", "--no-kernel-timing", "--no-other-configs"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).print_callers("method 'to' of")
print(s.getvalue()[:5000])
