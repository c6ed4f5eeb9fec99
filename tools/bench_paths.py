#!/usr/bin/env python3
"""Throughput of the paths that are NOT on the bench line: other ladders, formats, engines and options,
inputs resident in HBM (noise + bursts of one synthetic receiver replicated), chain up to its device outputs.
usage: tools/bench_paths.py [R]   -> one line per configuration: IQ MS/s, ms per step
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from ais_catcher_amd import gpu, synth, workload  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402  (the parity gate of the bench line, shared)


GATE = {"checked": 0, "failed": []}


def run(name, R, rate, block, fmt="cf32", steps=30, **kw):
    """One path: 30 untimed + `steps` timed steps over two resident blocks (one synthetic receiver replicated R times; with
    BENCH_PATHS_DISTINCT=1 and CF32 input the bench line's batch of R DISTINCT receivers, workload.resident_batch), then the
    SAME parity gate as bench.py on the run that was just timed: the last block's outputs of the first and the last receiver (eight receivers
    spread over the batch when they are distinct) against the oracle fed the same block sequence, bit for bit (hard bits, levels, ppm, FM signs, 48 kHz channels -- whatever
    the engine hands out)."""
    model = kw.get("model", gpu.MODEL_DEFAULT)
    only = os.environ.get("BENCH_PATHS_ONLY")
    if only and only not in name:
        return
    distinct = bool(os.environ.get("BENCH_PATHS_DISTINCT")) and fmt == "cf32" and not kw.get("mode_x")
    x = synth.receiver_stream(block * 2, sample_rate=rate, receiver_id=7, single_channel=kw.get("mode_x", False))
    if fmt == "cf32":
        hostx = x
        host = x.view(np.float32).reshape(2, block, 2)
        code = gpu.FMT_CF32
    elif fmt == "cu8":
        hostx = synth.to_cu8(x)
        host = hostx.reshape(2, block, 2)
        code = gpu.FMT_CU8
    else:
        hostx = synth.to_cs16(x)
        host = hostx.reshape(2, block, 2)
        code = gpu.FMT_CS16
    per = len(hostx) // 2
    dev = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    data = dev.unsqueeze(1).expand(2, R, block, 2).contiguous()   # [2 blocks][R][block][2]
    if distinct:
        del data
        data = workload.resident_batch(torch, R, 2, block=block, sample_rate=rate)
        name += " [distinct receivers]"
    torch.cuda.synchronize()
    g = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=block, input_format=code, **kw)
    warm = 30
    for i in range(warm):  # warm-up incl. the clock ramp after the idle set-up time
        g.submit_device(data[i & 1].data_ptr(), block)
        g.run()
    g.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        g.submit_device(data[i & 1].data_ptr(), block)
        g.run()
    g.sync()
    dt = time.perf_counter() - t0
    verdict = "parity off"
    if not os.environ.get("BENCH_PATHS_NO_GATE"):
        try:
            g.sync_outputs()
        except gpu.AisGpuError as e:  # (device decoders: a throughput run never collects its frames, the decisions are copied all the same)
            if not (kw.get("gpu_decode") and "(5)" in str(e)):
                raise
        seq = [i & 1 for i in range(warm)] + [i & 1 for i in range(steps)]
        frames = None
        if kw.get("gpu_decode"):  # one more block behind the timed ones: its frames alone are in the ring (bench.py does the same)
            seq.append(steps & 1)
            g.submit_device(data[steps & 1].data_ptr(), block)
            g.run()
            g.sync_outputs()
            frames = g.frames()
        okw = {k: kw[k] for k in ("ps_ema", "fp_ds", "mode_x", "dsk", "ma") if k in kw}
        blocks = [hostx[:per], hostx[per:]]
        blocks_of = lambda r: blocks
        if distinct:
            blocks_of = lambda r: [np.ascontiguousarray(data[b, r].cpu().numpy()).view(np.complex64).reshape(-1) for b in range(2)]
        # replicated receivers are all the same stream: the first and the last say everything; distinct ones: eight spread over the batch
        gated = sorted({0, R - 1}) if not distinct else sorted(set(int(round(i * (R - 1) / 7.0)) for i in range(8)))
        n, bad = bench.parity_check(g, None, seq, gated, rate=rate, model=model, fmt=fmt, blocks_of=blocks_of, frames=frames, **okw)
        GATE["checked"] += n
        verdict = ("parity: %d receivers bit-exact%s" % (n, " incl. the NMEA text of %d device frames" % len(frames) if frames is not None else "")) if not bad else "PARITY MISMATCH: " + "; ".join(bad[:4])
        if bad:
            GATE["failed"].append(name)
    g.close()
    del data, dev
    print("%-64s %9.0f MS/s   %7.3f ms per step of %d x %d samples   %s" % (name, R * block * steps / dt / 1e6, dt / steps * 1e3, R, block, verdict), flush=True)


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = 786432
    run("ModelDefault 1536k CF32 (the bench line)", R, 1536000, B)
    run("ModelDefault 1536k CF32, frame decoders on the device (K7e)", R, 1536000, B, gpu_decode=True)
    run("ModelDefault 1536k CF32, PS_EMA off (boxcar PhaseSearch, chunk-parallel)", R, 1536000, B, ps_ema=False)
    run("ModelDefault 1536k CU8", R, 1536000, B, fmt="cu8")
    run("ModelDefault 1536k CU8, FP_DS (fixed-point ladder)", R, 1536000, B, fmt="cu8", fp_ds=True)
    run("ModelDefault 1536k CS16", R, 1536000, B, fmt="cs16")
    run("ModelDefault 768k CF32 (three CIC5 stages)", R, 768000, B // 2)
    run("ModelDefault 3072k CF32 (five CIC5 stages in the front-end waves)", R, 3072000, B)
    run("ModelDefault 3072k CU8", R, 3072000, B, fmt="cu8")
    run("ModelDefault 6144k CF32 (six CIC5 stages in the front-end waves)", R // 2, 6144000, B * 2)
    run("ModelDefault 12288k CF32 (pre-decimation pass of three stages)", R // 2, 12288000, B * 2)
    run("ModelDefault 10 MSPS CF32 (Airspy R2: pre-decimation pass of five stages + resampler K1u)", R // 2, 10000000, B * 2)
    run("ModelDefault 10 MSPS CU8", R // 2, 10000000, B * 2, fmt="cu8")
    run("ModelDefault 6 MSPS CF32 (pre-decimation + resampler K1u)", R, 6000000, B)
    # (the low-rate ladders with as many receivers as make a step of the bench line's size, 1.6 GB of input: 256 receivers are a
    # 0.4 / 0.1 GB step there, which the latency of the kernel chain bounds, not any kernel)
    run("ModelDefault 288k CF32 (decimate-by-3 front end K1k)", R * 4, 288000, 49152 * 4)
    run("ModelDefault 250k CF32 (resampled into 288k: Upsample in front of DownsampleKFilter)", R * 4, 250000, 49152 * 4)
    run("ModelDefault 2400k CF32 (resampled into 3072k)", R, 2400000, B)
    run("ModelDefault 96k CF32 (dual channel, the ladder's last bucket)", R * 16, 96000, 1024 * 48)
    run("ModelDefault mode X 96k CF32 (single channel K1x)", R * 16, 96000, 1024 * 48, mode_x=True)
    run("ModelDefault mode X 48k CF32", R * 32, 48000, 512 * 48, mode_x=True)
    run("ModelDefault mode X 192k CF32", R * 8, 192000, 2048 * 48, mode_x=True)
    run("ModelChallenger 1536k CF32 (fused back end, FM branch inside the derotation / FIR kernel)", R, 1536000, B, model=gpu.MODEL_CHALLENGER)
    run("ModelChallenger 1536k CF32, twenty decoders on the device", R, 1536000, B, model=gpu.MODEL_CHALLENGER, gpu_decode=True)
    run("ModelChallenger 6 MSPS CF32 (BASELINE configs[2])", R, 6000000, B, model=gpu.MODEL_CHALLENGER)
    run("ModelBase 1536k CF32 (front end + FM receiver, signs out)", R, 1536000, B, model=gpu.MODEL_BASE)
    run("ModelBase 1536k CF32, SimplePLL + decoder on the device", R, 1536000, B, model=gpu.MODEL_BASE, gpu_decode=True)
    run("ModelStandard 1536k CF32, five decoders on the device", R, 1536000, B, model=gpu.MODEL_STANDARD, gpu_decode=True)
    run("ModelEngineV2 1536k CF32 (front end + estimates / energies / FM branch, c48 to the host)", R, 1536000, B, model=gpu.MODEL_V2)
    run("ModelEngineV2 1536k CF32, the whole engine on the device (kv2_engine; frames out, no 48 kHz channels over PCIe)", R, 1536000, B, model=gpu.MODEL_V2, gpu_decode=True)
    print("parity gate: %d receiver outputs compared, %d paths failed %s" % (GATE["checked"], len(GATE["failed"]), GATE["failed"]))
    sys.exit(3 if GATE["failed"] else 0)
