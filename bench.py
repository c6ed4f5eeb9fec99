#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through the ModelDefault demodulation chain on MI355X.

Workload (BASELINE.json configs[3] / configs[4]): 256 batched dual-channel receivers per GPU,
1536 kSPS CFLOAT32, one reference Receive() block (786,432 IQ samples) per receiver per step,
synthetic GMSK bursts + AWGN, inputs resident in HBM before the timed region.  One process per GPU;
receivers are independent, so N GPUs run N x 256 receivers with no collective (weak scaling).

Prints ONE JSON line (rank 0): metric/value/unit..., plus
  roofline     -- front-end kernel (the HBM-bound kernel): algorithmic bytes per launch
                  (8.30 B/IQ sample) / its average launch time measured with HIP events
                  on the library's own stream
  cpu_baseline -- the reference's own sources compiled with its shipped flags (oracle/_ref), timed on
                  this host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

ALGO_BYTES_PER_SAMPLE = 8.30   # 8 B read + 0.25 B hard bits + 0.05 B level
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
RATE = 1536000
BLOCK = 786432


def make_resident_input(torch, n_rx, n_blocks, seed, unique=8):
    """[n_blocks][n_rx][BLOCK] complex64 on the GPU: `unique` CPU-synthesised burst streams, shared by
    receiver groups with a per-receiver slot shift, plus independent AWGN per receiver (torch RNG)."""
    import _pkg
    _pkg.load()
    from ais_catcher_amd import synth
    slot = 40960
    base = []
    for u in range(unique):
        x = synth.receiver_stream(BLOCK * n_blocks, receiver_id=seed * 1000 + u, noise_sigma=0.0)
        base.append(torch.from_numpy(x.view(np.float32).reshape(n_blocks, BLOCK, 2)))
    base = torch.stack(base).cuda()                      # [unique][n_blocks][BLOCK][2]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(12345 + seed)
    out = torch.empty((n_blocks, n_rx, BLOCK, 2), dtype=torch.float32, device="cuda")
    for r in range(n_rx):
        b = torch.roll(base[r % unique], shifts=(r // unique) * slot, dims=1)
        out[:, r] = b
    out.add_(torch.randn(out.shape, generator=gen, device="cuda", dtype=torch.float32), alpha=0.01)
    torch.cuda.synchronize()
    return out


def cpu_baseline(seconds=12.0):
    """Reference chain (shipped flags -O3 -ffast-math) on the host cores, one ModelDefault per thread."""
    import checkers
    import _pkg
    _pkg.load()
    from ais_catcher_amd import synth
    if checkers.have_ref("fast"):
        kind, mk = "reference", (lambda: checkers.Ref(model=2, rate=RATE, fmt="cf32", kind="fast"))
    else:
        kind, mk = "port", (lambda: checkers.Oracle(model=2, rate=RATE, fmt="cf32"))
    cores = max(1, min(os.cpu_count() or 1, 64))
    nblk = 4
    x = synth.receiver_stream(BLOCK * nblk, receiver_id=4242)
    blocks = [np.ascontiguousarray(x[i * BLOCK:(i + 1) * BLOCK]) for i in range(nblk)]
    chains = [mk() for _ in range(cores)]
    counts = [0] * cores
    t_end = time.perf_counter() + seconds

    def work(i):
        c = chains[i]
        k = 0
        while time.perf_counter() < t_end:
            c.feed(blocks[k % nblk])
            k += 1
        counts[i] = k

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(counts) * BLOCK
    return {"value": round(total / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": kind,
            "sample": "%d blocks of %d CF32 IQ samples over %d threads in %.1f s (4 distinct blocks cycled, "
                      "one ModelDefault instance per thread, in-memory)" % (sum(counts), BLOCK, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--receivers", type=int, default=256)
    ap.add_argument("--preroll", type=int, default=40, help="untimed steps before the warm-up (GPU clock ramp)")
    ap.add_argument("--gpu-decode", action="store_true", help="also run the AIS::Decoder state machines on the device (frames out)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import _pkg
    _pkg.load()
    from ais_catcher_amd import gpu, shard

    R = args.receivers
    nb = 2  # distinct resident blocks cycled (2 x 1.6 GB)
    rx_ids = shard.receiver_range(rank, world, R)  # this rank's receivers; no other rank touches them
    data = make_resident_input(torch, R, nb, seed=rx_ids[0] // R)
    g = gpu.AisGpu(sample_rate=RATE, n_receivers=R, block_len=BLOCK, device_id=local, gpu_decode=args.gpu_decode)

    def step(i):
        g.submit_device(data[i % nb].data_ptr(), BLOCK)
        g.run()

    # Clock ramp: after the idle time of the set-up the GPU runs its first ~8 ms of load at a low clock and then pauses
    # for ~6 ms while it switches up (tools/host_times.py: run() call 5 of a cold context blocks for 7.5 ms).  A few
    # dozen untimed steps in front of the W warm-up steps keep that one-off transient out of the K timed steps.
    for i in range(args.preroll):
        step(i)
    g.sync()
    for i in range(args.warmup):
        step(i)
    g.sync()
    g.timing(not os.environ.get("BENCH_NO_K1_EVENTS"))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    t_enq = time.perf_counter() - t0   # host time to enqueue all steps (no device sync inside)
    g.sync()
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dist, device="cuda")

    k1_ms, k1_n = g.frontend_ms()
    g.close()
    # the same kernel measured without the other streams' kernels competing for the chip (untimed extra steps)
    gs = gpu.AisGpu(sample_rate=RATE, n_receivers=R, block_len=BLOCK, device_id=local, serial=True)
    for i in range(2):
        gs.submit_device(data[i % nb].data_ptr(), BLOCK)
        gs.run()
    gs.sync()
    gs.timing(True)
    for i in range(4):
        gs.submit_device(data[i % nb].data_ptr(), BLOCK)
        gs.run()
    iso_ms, _ = gs.frontend_ms()
    gs.close()
    samples_per_step = R * BLOCK
    value = shard.aggregate_msamples(samples_per_step, world, args.steps, dt)
    achieved = samples_per_step * ALGO_BYTES_PER_SAMPLE / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    # HBM bytes per launch of the same kernel from rocprofv3 PMC passes (FETCH_SIZE/WRITE_SIZE, collected separately
    # by tools/pmc_traffic.sh on this workload and committed under profiles/); expressed like `achieved` (GB/s)
    traffic = traffic_bytes = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic_k1.json")
    if os.path.exists(tpath) and R == 256 and k1_ms > 0:
        traffic_bytes = json.load(open(tpath))["hbm_bytes_per_launch"]
        traffic = round(traffic_bytes / (k1_ms * 1e-3) / 1e9, 1)
    res = {
        "metric": "IQ Msamples/s (CFLOAT32) through ModelDefault chain", "value": round(value, 1),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %d batched dual-channel receivers per GPU, 1536 kSPS CF32, "
                               "%d IQ samples per receiver per step, resident in HBM, chain up to hard bits/levels/ppm"
                               % (R, BLOCK),
                   "gpu_frame_decoder": bool(args.gpu_decode), "receivers_per_gpu": R, "block_len": BLOCK, "sample_rate": RATE, "parallelism": "receivers sharded, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_bytes_per_launch": traffic_bytes,
                     "algorithmic_bytes_per_launch": samples_per_step * ALGO_BYTES_PER_SAMPLE,
                     "kernel": "k1_dpp (front end: CIC5 ladder + FDC + Rotate + DS2 + FCIC5 + the spectral analysis of every window: "
                               "FFT-512, prefix sum, peak searches)", "avg_launch_ms": round(k1_ms, 4), "launches": k1_n,
                     # the same algorithmic bytes over the whole step (all kernels of the chain, wall clock / steps)
                     "whole_chain_frac": round(samples_per_step * ALGO_BYTES_PER_SAMPLE / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "isolated_launch_ms": round(iso_ms, 4),
                     "isolated_frac": round(samples_per_step * ALGO_BYTES_PER_SAMPLE / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if iso_ms > 0 else None},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
