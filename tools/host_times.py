import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, _pkg
_pkg.load()
from ais_catcher_amd import gpu
R, BLOCK = 256, 786432
x = torch.zeros((R, BLOCK, 2), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
g = gpu.AisGpu(sample_rate=1536000, n_receivers=R, block_len=BLOCK)
ts = []
t00 = time.perf_counter()
for i in range(30):
    t0 = time.perf_counter()
    g.submit_device(x.data_ptr(), BLOCK); g.run()
    ts.append((time.perf_counter() - t0) * 1e3)
g.sync()
print("total", (time.perf_counter() - t00) * 1e3)
print(" ".join("%.2f" % v for v in ts))
