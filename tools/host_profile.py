import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class A: pass
a = A(); a.n_env=1024; a.n_agent=3; a.horizon=25; a.minibatch=4096; a.repeat=1; a.dispatch="per_agent"
env, net, algo, buf, col = bench.build_job(a, torch.device("cuda"), 0)
for _ in range(10): bench.one_step(a, algo, buf, col)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): bench.one_step(a, algo, buf, col)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
