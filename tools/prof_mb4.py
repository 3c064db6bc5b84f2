import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.scale_smoke as S
t0=time.perf_counter(); p=torch.randperm(819200); print("torch.randperm(819200) first: %.1f ms" % ((time.perf_counter()-t0)*1e3))
t0=time.perf_counter(); p=torch.randperm(819200); print("torch.randperm(819200) second: %.1f ms" % ((time.perf_counter()-t0)*1e3))
t0=time.perf_counter(); q=p.to("cuda:0"); torch.cuda.synchronize(); print("H2D 6.5 MB: %.2f ms" % ((time.perf_counter()-t0)*1e3))
print("cpu threads", torch.get_num_threads(), os.cpu_count())
pr = cProfile.Profile(); pr.enable()
S.run("mpe_disc_mb4", S.CASES["mpe_disc_mb4"])
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000])
