import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _pkg; _pkg.load()
from ais_catcher_amd import gpu, synth
import checkers
# 1536k ladder with a block whose group count (2457) ends in a chunk of 409 symbols
rate, block = 1536000, 32 * 12288
x = synth.receiver_stream(block * 2, sample_rate=rate, receiver_id=54, gap_slots=(0, 1))
g = gpu.AisGpu(sample_rate=rate, n_receivers=1, block_len=block, taps=False)
o = checkers.Oracle(model=2, rate=rate, fmt="cf32", taps=True)
o.feed_blocks(x, block)
gd = 0
for b in range(2):
    g.submit(0, x[b * block:(b + 1) * block]); g.run(); g.sync_outputs()
    out = g.fetch(0, 0, 0)
    n = out["n_groups"]
    for j in range(5):
        ob = o.bits(0, j)[0][gd:gd + n]
        bad = np.nonzero(out["bits"][j] != ob)[0]
        print("blk", b, "phase", j, "n", n, "mismatches", len(bad), bad[:5])
    gd += n
