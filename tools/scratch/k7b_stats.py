import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, _pkg
_pkg.load()
from ais_catcher_amd import gpu, synth, workload
B = 786432
for distinct in (1,):
    if distinct:
        data = workload.resident_batch(torch, 256, 2)
    else:
        x = synth.receiver_stream(B * 2, receiver_id=7)
        dev = torch.from_numpy(np.ascontiguousarray(x.view(np.float32).reshape(2, B, 2))).cuda()
        data = dev.unsqueeze(1).expand(2, 256, B, 2).contiguous()
    g = gpu.AisGpu(sample_rate=1536000, n_receivers=256, block_len=B, model=gpu.MODEL_BASE, gpu_decode=True)
    for i in range(1):
        g.submit_device(data[i & 1].data_ptr(), B); g.run()
        g.sync()
        print("distinct", distinct, "block", i, flush=True)
        try:
            g.sync_outputs()
        except Exception as e:
            print("sync_outputs:", e)
    g.close()
