#!/usr/bin/env python3
"""Side measurement (not the bench contract): the bench workload with integer input formats resident in HBM --
CU8 through the float ladder, CU8 through the fixed-point ladder (`-go FP_DS on`), CS8, CS16 -- next to CF32.
usage: python tools/bench_formats.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R, BLOCK, RATE = 256, 786432, 1536000


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import torch
    import _pkg
    _pkg.load()
    from ais_catcher_amd import gpu
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    cases = [("cf32", gpu.FMT_CF32, 8, {}), ("cu8", gpu.FMT_CU8, 2, {}), ("cu8 fp_ds", gpu.FMT_CU8, 2, {"fp_ds": True}),
             ("cs8", gpu.FMT_CS8, 2, {}), ("cs16", gpu.FMT_CS16, 4, {})]
    for name, fmt, nbytes, kw in cases:
        data = []
        for b in range(2):
            if fmt == gpu.FMT_CF32:
                data.append((torch.randn(R, BLOCK * 2, device=dev, generator=gen) * 0.05).contiguous())
            else:
                raw = torch.randint(96, 160, (R, BLOCK * nbytes), device=dev, generator=gen, dtype=torch.int16).to(torch.uint8)
                data.append(raw.contiguous())
        g = gpu.AisGpu(sample_rate=RATE, n_receivers=R, block_len=BLOCK, input_format=fmt, **kw)
        for i in range(40):
            g.submit_device(data[i % 2].data_ptr(), BLOCK)
            g.run()
        g.sync()
        g.timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g.submit_device(data[i % 2].data_ptr(), BLOCK)
            g.run()
        g.sync()
        dt = time.perf_counter() - t0
        k1_ms, n = g.frontend_ms()
        g.close()
        algo = nbytes + 0.30
        print(json.dumps({"format": name, "ms_per_step": round(dt / steps * 1e3, 4), "Msamples_per_s": round(R * BLOCK * steps / dt / 1e6, 1),
                          "k1_avg_launch_ms": round(k1_ms, 4), "algorithmic_bytes_per_sample": algo,
                          "k1_algorithmic_GBps": round(R * BLOCK * algo / (k1_ms * 1e-3) / 1e9, 1) if k1_ms > 0 else None}), flush=True)
        del data
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
