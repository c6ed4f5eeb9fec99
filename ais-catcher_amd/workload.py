"""The batch workload of BASELINE.json configs[3] / [4]: R dual-channel receivers at 1536 kSPS CF32, one
reference Receive() block (786,432 IQ samples) per receiver per step, synthetic GMSK bursts + AWGN, resident in HBM.

Shared by bench.py (the measured input) and tests/test_gpu_parity.py (the same input checked against the oracle), so that
what is timed is what is verified.  torch is used for device memory and its RNG only; nothing here touches the library.
"""
import numpy as np

from . import synth

RATE = 1536000
BLOCK = 786432   # the reference file reader's CF32 block: 24 * 16 * 16384 B / 8 (Device/FileRAW.h:43)
SLOT = 40960     # samples per AIS slot at 1536 kSPS (256 bits x 160 samples)


def resident_batch(torch, n_rx, n_blocks, seed=0, unique=8, block=BLOCK, device="cuda", sample_rate=RATE):
    """-> float32 tensor [n_blocks][n_rx][block][2] on `device`: `unique` CPU-synthesised burst streams (seeded), every
    receiver a slot-shifted copy of one of them plus its OWN white noise (sigma 0.01 per component, seeded device RNG):
    n_rx distinct streams for the price of `unique` runs of the modulator.  sample_rate: 1536000 (configs[1], [3], [4]) or
    6000000 (configs[2]); the roll is a whole number of slots at either rate."""
    base = []
    for u in range(unique):
        x = synth.receiver_stream(block * n_blocks, sample_rate=sample_rate, receiver_id=seed * 1000 + u, noise_sigma=0.0)
        base.append(torch.from_numpy(x.view(np.float32).reshape(n_blocks, block, 2)))
    base = torch.stack(base).to(device)                      # [unique][n_blocks][block][2]
    gen = torch.Generator(device=device)
    gen.manual_seed(12345 + seed)
    out = torch.empty((n_blocks, n_rx, block, 2), dtype=torch.float32, device=device)
    for r in range(n_rx):
        out[:, r] = torch.roll(base[r % unique], shifts=(r // unique) * (SLOT * (sample_rate // 1000) // (RATE // 1000)), dims=1)
    out.add_(torch.randn(out.shape, generator=gen, device=device, dtype=torch.float32), alpha=0.01)
    if device != "cpu":
        torch.cuda.synchronize()
    return out


def block_sequence(preroll, warmup, steps, n_blocks):
    """Index of the resident block every step of a bench run consumes, in order (pre-roll, warm-up, timed steps)."""
    return [i % n_blocks for i in range(preroll)] + [i % n_blocks for i in range(warmup)] + \
           [(warmup + i) % n_blocks for i in range(steps)]


def host_stream(batch, rx, sequence):
    """The CF32 stream receiver `rx` saw over `sequence` (block indices), as one complex64 array on the host."""
    blocks = {b: batch[b, rx].cpu().numpy().reshape(-1).view(np.complex64) for b in set(sequence)}
    return np.concatenate([blocks[b] for b in sequence])
