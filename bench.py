#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s through the ModelDefault demodulation chain on MI355X.

Default workload (BASELINE.json configs[3] / configs[4], `--config 4`): 256 batched dual-channel receivers per GPU,
1536 kSPS CFLOAT32, one reference Receive() block (786,432 IQ samples) per receiver per step,
synthetic GMSK bursts + AWGN, inputs resident in HBM before the timed region.  One process per GPU;
receivers are independent, so N GPUs run N x 256 receivers with no collective (weak scaling).
`--config 2` = configs[1] (ONE receiver: the latency case), `--config 3` = configs[2] (6 MSPS AirSpy-rate input, deeper CIC5
ladder + fractional resampler, ModelChallenger: 8.14 algorithmic B/sample) -- timed and parity-gated the same way.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4]

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this script starts the N ranks itself (one
process per GPU, LOCAL_RANK = device); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
it is one of the ranks.  The ranks only meet in a timing barrier and the max-over-ranks of the elapsed time
(torch.distributed, gloo: there is no RCCL traffic on this path -- SURVEY.md 8(e)).

Prints ONE JSON line (rank 0): metric/value/unit..., plus
  roofline       -- front-end kernel (the HBM-bound kernel): algorithmic bytes per launch
                    (8.30 B/IQ sample) / its average launch time measured with HIP events
                    on the library's own stream over the timed region
  cpu_baseline   -- the reference's own sources compiled with its shipped flags (oracle/_ref), timed on
                    this host's cores on a bounded sample of the same workload (rank 0, N=1 only)
  parity_checked -- receivers whose hard bits / levels / ppm of the LAST TIMED block were compared bit for
                    bit with the oracle fed the same block sequence (SURVEY.md 8(d) "parity gates in the
                    same run"); a mismatch makes the run fail (exit status 3)
  other_configs  -- (default workload, N=1) BASELINE configs[1] and configs[2] timed and parity-gated by this same script
                    behind the main timed region (20 steps each, own processes): value, ms_per_step, whole_chain_frac
"""
import argparse
import json
import os
import socket
import subprocess
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

# BASELINE.json configs, numbered as SURVEY.md does (config 1 = the CPU-only plumbing case, config 5 = config 4 on 8 GPUs)
CONFIGS = {
    2: {"key": "configs[1]", "rate": 1536000, "model": 2, "model_name": "ModelDefault", "receivers": 1, "bytes": 8.30,
        "kernel": "k1_dpp (front end incl. the spectral analysis)",
        "what": "ONE dual-channel receiver, 1536 kSPS CF32: latency of a block through the chain (the batch of one is launch / recurrence bound)"},
    3: {"key": "configs[2]", "rate": 6000000, "model": 4, "model_name": "ModelChallenger", "receivers": 256, "bytes": 8.14,
        "kernel": "k1_dpp<4, CF32, PRE> (the four CIC5 stages at the input rate: the pass that reads the input stream)",
        "what": "%d batched dual-channel receivers per GPU, 6 MSPS CF32 (AirSpy rate): 4 x CIC5 -> Upsample 125/128 -> 2 x CIC5 -> ... -> "
                "coherent chain + FM branch (ModelChallenger)"},
    4: {"key": "configs[3]", "rate": 1536000, "model": 2, "model_name": "ModelDefault", "receivers": 256, "bytes": 8.30,
        "kernel": "k1_dpp (front end: CIC5 ladder + FDC + Rotate + DS2 + FCIC5 + the spectral analysis of every window: FFT-512, prefix sum, peak searches)",
        "what": "%d batched dual-channel receivers per GPU, 1536 kSPS CF32"},
    # not a BASELINE config: configs[3]'s batch with the input as CU8 (what BASELINE configs[0] reads and RTL-SDRs deliver), converted
    # inside the front-end kernel (Utilities/Convert.cpp:255-264).  2.30 algorithmic B/sample: the same arithmetic on 3.6 x fewer bytes,
    # so its step time is the chain's compute / latency floor -- reported in other_configs of the default run
    5: {"key": "configs[3] with CU8 input", "rate": 1536000, "model": 2, "model_name": "ModelDefault", "receivers": 256, "bytes": 2.30, "format": "cu8",
        "kernel": "k1_dpp<4, CU8, false> (the same front end, CU8 -> CF32 in its lanes)",
        "what": "%d batched dual-channel receivers per GPU, 1536 kSPS CU8 (the synthetic CF32 batch as round(x * 128 + 128) bytes)"},
}


def host_description():
    """nproc / lscpu of the machine the CPU baseline runs on (SURVEY.md 8(d) asks for it in the report)."""
    info = {"nproc": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for line in txt.splitlines():
            if ":" in line:
                k, v = line.split(":", 1)
                kv[k.strip()] = v.strip()
        info["model"] = kv.get("Model name")
        sockets, cps, tpc = kv.get("Socket(s)"), kv.get("Core(s) per socket"), kv.get("Thread(s) per core")
        if sockets and cps:
            info["physical_cores"] = int(sockets) * int(cps)
        if tpc:
            info["threads_per_core"] = int(tpc)
    except Exception:
        pass
    try:
        info["usable_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return info


def cpu_topology():
    """[(cpu, package, core_id, numa_node)] of the CPUs this process may run on, from /sys (no numactl / hwloc in the image)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    node_of = {}
    nd = "/sys/devices/system/node"
    if os.path.isdir(nd):
        for n in os.listdir(nd):
            if n.startswith("node") and n[4:].isdigit():
                for c in os.listdir(os.path.join(nd, n)):
                    if c.startswith("cpu") and c[3:].isdigit():
                        node_of[int(c[3:])] = int(n[4:])
    topo = []
    for c in allowed:
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except Exception:
            pkg, core = 0, c
        topo.append((c, pkg, core, node_of.get(c, 0)))
    return topo


def pick_cpus(topo, n):
    """n CPUs, one hardware thread per physical core first, spread round robin over the NUMA nodes."""
    by_node = {}
    seen = set()
    second = []
    for c, pkg, core, node in topo:
        if (pkg, core) in seen:
            second.append((c, node))
            continue
        seen.add((pkg, core))
        by_node.setdefault(node, []).append(c)
    order = []
    lists = [by_node[k] for k in sorted(by_node)]
    k = 0
    while any(lists):
        lst = lists[k % len(lists)]
        if lst:
            order.append(lst.pop(0))
        k += 1
    order += [c for c, _ in second]
    return order[:n]


def cgroup_cpu_limit():
    """CPU bandwidth limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), None if unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def cgroup_throttled_usec():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if k in ("throttled_usec", "throttled_time"):
                    return int(v) / (1 if k == "throttled_usec" else 1000)
        except Exception:
            pass
    return None


def cpu_baseline(seconds=16.0, model=2, rate=RATE):
    """Reference chain (shipped flags -O3 -ffast-math), one independent ModelDefault per thread.  The timing loop lives in the
    checker library (oracle/ref_harness.cpp: ref_bench_threads): every thread is pinned to a physical core of its own (spread
    over the NUMA nodes), builds its model and copies its input blocks itself (first touch on its own node), and all threads
    start together.  Reported: a 1-thread figure, a short scan over thread counts, and the long run at the best count --
    per core and aggregate -- next to the container's cgroup CPU limit, which on a shared host decides how many
    of the visible cores actually run."""
    import ctypes
    import checkers
    import _pkg
    _pkg.load()
    from ais_catcher_amd import synth
    host = host_description()
    topo = cpu_topology()
    phys = len(set((p, c) for _, p, c, _ in topo))
    host["usable_cpus"] = len(topo)
    host["usable_physical_cores"] = phys
    host["numa_nodes"] = len(set(n for _, _, _, n in topo))
    host["cgroup_cpu_limit_cores"] = cgroup_cpu_limit()
    nblk = 4
    x = synth.receiver_stream(BLOCK * nblk, sample_rate=rate, receiver_id=4242)
    if not checkers.have_ref("fast"):
        return cpu_baseline_port(seconds, x, nblk, host, phys, model, rate)
    lib = ctypes.CDLL(os.path.join(checkers.ORACLE_DIR, "_ref", "libaisref_fast.so"))
    lib.ref_bench_threads.restype = ctypes.c_double
    lib.ref_bench_threads.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    buf = np.ascontiguousarray(x[:BLOCK * nblk]).astype(np.complex64)

    def run(n, secs):
        cpus = np.array(pick_cpus(topo, n), np.int32)
        n = len(cpus)
        counts = np.zeros(n, np.int64)
        dt = lib.ref_bench_threads(model, rate, 1, buf.ctypes.data, nblk, BLOCK * 8, n, cpus.ctypes.data, secs, counts.ctypes.data)
        return n, float(counts.sum()) * BLOCK / dt / 1e6, dt, int(counts.sum())

    thr0 = cgroup_throttled_usec()
    _, single, _, _ = run(1, 1.5)
    limit = host["cgroup_cpu_limit_cores"]
    # thread counts up to what the container may actually run: above the cgroup quota every further thread only adds throttling
    # (one oversubscribed point is kept in the scan to show it, but is never the reported figure)
    cap = min(len(topo), int(limit)) if limit and limit >= 1 else len(topo)
    cand = sorted(set(max(1, c) for c in [phys // 8, phys // 4, phys // 2, phys, cap // 2, cap] if c <= cap))
    scan, over = {}, {}
    for n in cand:
        nn, v, _, _ = run(n, 1.0)
        scan[nn] = round(v, 1)
    if cap < len(topo):
        nn, v, _, _ = run(min(len(topo), 2 * cap), 1.0)
        over[nn] = round(v, 1)
    best = max(scan, key=lambda k: scan[k])
    n, value, dt, blocks = run(best, max(4.0, seconds - 1.5 - 1.0 * len(cand)))
    thr1 = cgroup_throttled_usec()
    return {"value": round(value, 2), "unit": "Msamples/s", "cores": n, "kind": "reference",
            "per_thread": round(value / n, 2), "single_thread": round(single, 2), "thread_scan": scan, "thread_scan_over_quota": over,
            "cgroup_throttled_s": None if thr0 is None or thr1 is None else round((thr1 - thr0) / 1e6, 2), "host": host,
            "sample": "%d blocks of %d CF32 IQ samples over %d pinned threads (one per physical core, spread over %d NUMA nodes; the thread "
                      "count <= the cgroup quota with the highest aggregate of the scan) in %.1f s; %d distinct blocks cycled, one ModelDefault instance per "
                      "thread built and fed inside its thread, in-memory" % (blocks, BLOCK, n, host["numa_nodes"], dt, nblk),
            "model": model, "sample_rate": rate,
            "note": "the container's cgroup CPU limit (host.cgroup_cpu_limit_cores), not the host's core count, bounds the aggregate; "
                    "per_thread x physical cores would be an extrapolation, not a measurement"}


def cpu_baseline_port(seconds, x, nblk, host, phys, model=2, rate=RATE):
    """Fallback where the compiled reference is not there: the oracle's C restatement, one chain per Python thread (ctypes releases the GIL)."""
    import checkers
    cores = max(1, phys)
    blocks = [np.ascontiguousarray(x[i * BLOCK:(i + 1) * BLOCK]) for i in range(nblk)]
    chains = [checkers.Oracle(model=model, rate=rate, fmt="cf32") for _ in range(cores)]
    counts = [0] * cores
    t_end = time.perf_counter() + seconds

    def work(i):
        k = 0
        while time.perf_counter() < t_end:
            chains[i].feed(blocks[k % nblk])
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
    return {"value": round(total / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "per_thread": round(total / dt / 1e6 / cores, 2), "host": host,
            "sample": "%d blocks of %d CF32 IQ samples over %d threads in %.1f s (oracle restatement)" % (sum(counts), BLOCK, cores, dt)}


def parity_check(g, data, sequence, receivers, rate=RATE, model=2, fmt="cf32", blocks_of=None, frames=None, **okw):
    """Outputs of the LAST block of `sequence` (what the context `g` holds after sync_outputs) against the oracle fed the same
    sequence of resident blocks, for the given receivers: hard bits of all five sampling phases, levels, ppm -- for
    ModelChallenger also the sign of every filtered FM-discriminator sample (what its five FM decoders per channel see), for
    ModelBase / ModelStandard those signs alone (their whole device output), for ModelEngineV2 the 48 kHz channels.  A resampled
    rate completes one or two downstream blocks per input block: all of them are compared, in order.
    data: resident tensor [n_blocks][n_rx]... (CF32), or blocks_of(r) -> list of host arrays, one per resident block (any format).
    okw: options of the oracle chain (dsk, ps_ema, fp_ds, mode_x, ma).  Returns (receivers checked, list of mismatch descriptions)."""
    import checkers
    from concurrent.futures import ThreadPoolExecutor
    n_sub = g.out_count()

    def one(r):
        if blocks_of is not None:
            blocks = blocks_of(r)
        else:
            blocks = [data[b, r].cpu().numpy().reshape(-1).view(np.complex64) for b in range(data.shape[0])]
        o = checkers.Oracle(model=model, rate=rate, fmt=fmt, taps=True, **okw)
        o.set_taps(False)
        n_lines = 0
        for i, b in enumerate(sequence):
            if i == len(sequence) - 1:   # only the last block's outputs are compared (and recorded)
                o.set_taps(True)
                n_lines = len(o.nmea()) if frames is not None else 0
            o.feed(blocks[b])
        bad = []
        if frames is not None:   # --gpu-decode: the frames the device decoders completed in the last block, through the host tail
            # (validation, NMEA text), against the messages the oracle's decoders printed for that block -- in order; the multi-part
            # sentence id is a per-process counter: compared without it (and without the checksum over it)
            from ais_catcher_amd import host
            strip = lambda ls: [",".join(f for i, f in enumerate(l.split("*")[0].split(",")) if i != 3) for l in ls]
            hm = {2: host.ModelDefaultGPU, 4: host.ModelChallengerGPU, 0: host.ModelStandardGPU, 1: host.ModelBaseGPU, 11: host.ModelEngineV2GPU}[model](
                sample_rate=rate, detached=True, gpu_decode=True)
            for f in frames:
                if f["rx"] == r:
                    hm.frame(f)
            if strip(hm.nmea()) != strip(o.nmea()[n_lines:]):
                bad.append("rx %d: NMEA of the device decoders' frames (%d lines, the oracle printed %d)" % (r, len(hm.nmea()), len(o.nmea()) - n_lines))
            hm.close()
        for ch in range(1 if okw.get("mode_x") else 2):   # (channel mode X: one channel, the device's channel B stays silent)
            outs = [g.fetch(r, ch, s) for s in range(n_sub)]
            for a, b2 in zip(outs, outs[1:]):
                if b2["first_group"] != a["first_group"] + a["n_groups"] or b2["first_sample48"] != a["first_sample48"] + 512 * a["n_windows"]:
                    bad.append("rx %d ch %d: downstream blocks not contiguous" % (r, ch))
            if model in (2, 4):
                n_tot = sum(t["n_groups"] for t in outs)
                ol = None
                for j in range(5):
                    ob, ol_, oi = o.bits(ch, j)
                    ol = ol_
                    got = np.concatenate([t["bits"][j] for t in outs])
                    if len(ob) != n_tot or not np.array_equal(got, ob):
                        bad.append("rx %d ch %d phase %d: hard bits" % (r, ch, j))
                    elif n_tot and int(oi[0]) != 5 * outs[0]["first_group"] + j:
                        bad.append("rx %d ch %d phase %d: group bookkeeping" % (r, ch, j))
                if not np.array_equal(np.concatenate([t["lvl"] for t in outs]).view(np.uint32), ol.view(np.uint32)):
                    bad.append("rx %d ch %d: levels" % (r, ch))
                oppm = o.tap_ppm(2 + ch)
                gppm = np.concatenate([t["ppm"] for t in outs])
                if len(oppm) != len(gppm) or not np.array_equal(gppm.view(np.uint32), oppm.view(np.uint32)):
                    bad.append("rx %d ch %d: ppm" % (r, ch))
            if model in (4, 0, 1):   # FM receivers: Demod::FM -> Filter(Receiver); with Deinterleave(5) decoder j gets stream samples n = j (mod 5)
                fm = np.concatenate([t["fm_bits"] for t in outs])
                n0 = outs[0]["first_sample48"]
                idx = np.arange(n0, n0 + len(fm))
                for j in range(1 if model == 1 else 5):
                    of, _, _ = o.bits(ch, j, fm=1)
                    sel = fm if model == 1 else fm[idx % 5 == j]   # ModelBase: the sampler sees every sample
                    if len(of) != len(sel) or not np.array_equal(sel != 0, of > 0):
                        bad.append("rx %d ch %d FM stream %d: discriminator signs" % (r, ch, j))
            if model == 11 and frames is None:   # (the whole engine on the device: no 48 kHz channels come back, the frames above are the output)
                want = o.tap(ch)
                got = np.concatenate([t["c48"] for t in outs])
                if len(want) != len(got) or not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                    bad.append("rx %d ch %d: 48 kHz channel" % (r, ch))
        o.close()
        return bad

    with ThreadPoolExecutor(max_workers=min(len(receivers), os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, receivers))
    return len(receivers), [m for bad in res for m in bad]


def pmc_traffic_pass(config, receivers):
    """HBM-side bytes per launch of the front-end kernel from the PMC counters, measured in THIS session: two separate rocprofv3
    --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a short serial-mode run of
    this script.  FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced reads); both counters are in KiB.
    Returns a dict or None (no rocprofv3 on PATH, or a pass failed)."""
    import shutil
    import sqlite3
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    vals, per_kernel = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            env = dict(os.environ, AISGPU_SERIAL="1", TMPDIR="/tmp")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--config", str(config), "--receivers", str(receivers), "--steps", "3", "--warmup", "1", "--preroll", "2", "--no-other-configs",
                   "--no-cpu-baseline", "--parity-receivers", "0", "--no-pmc"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            db = None
            for root, _, files in os.walk(out):
                for f in files:
                    if f.endswith(".db"):
                        db = os.path.join(root, f)
            c = sqlite3.connect(db)
            r = c.execute("select sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%k1_dpp%' and counter_name=?",
                          (counter,)).fetchone()
            vals[counter] = r[0] / r[1]
            # the whole step: every kernel of the library (one launch of each per step; the derotation / FIR kernel's four template
            # instances take turns), summed and divided by the number of steps = launches of the front-end kernel
            per_kernel[counter] = {}
            for name, v in c.execute("select kernel_name, sum(value) from counters_collection where kernel_name like '%aisk::%' and counter_name=? "
                                     "group by kernel_name", (counter,)).fetchall():
                short = name.split("aisk::")[1].split("(")[0].split("<")[0]
                per_kernel[counter][short] = per_kernel[counter].get(short, 0.0) + v / r[1]
        step = {k: per_kernel["FETCH_SIZE"].get(k, 0.0) * 1024 * 2 + per_kernel["WRITE_SIZE"].get(k, 0.0) * 1024
                for k in set(per_kernel["FETCH_SIZE"]) | set(per_kernel["WRITE_SIZE"])}
        return {"fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                "hbm_bytes_per_launch": vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024, "host": socket.gethostname(),
                "step_bytes": sum(step.values()), "step_bytes_by_kernel": {k: round(v) for k, v in sorted(step.items(), key=lambda kv: -kv[1])},
                "step_read_bytes": sum(per_kernel["FETCH_SIZE"].values()) * 1024 * 2, "step_written_bytes": sum(per_kernel["WRITE_SIZE"].values()) * 1024,
                "note": "this session: separate rocprofv3 --pmc passes (serial mode, 3 launches each); FETCH_SIZE doubled per MI355X_MICROARCH.md"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_configs(steps=20, warmup=5):
    """BASELINE configs[1] (one receiver), configs[2] (6 MSPS, ModelChallenger, 256 receivers) and the default batch with CU8 input
    (2.30 algorithmic B/sample: the chain's compute floor, its whole_chain_frac priced at those bytes) behind the default workload's timed
    region and gate: the same code path (this script with --config 2 / 3, the driver's 20-step shape, parity gate on), each in a
    process of its own so that nothing of the main run is resident beside it.  One short record per config in the line.  A parity
    MISMATCH in one of them fails the whole run (main() returns 3, like a mismatch of the default workload); a child that times out or
    dies leaves an "error" / a non-zero "exit_status" in its record and the main line stands (its numbers do not depend on them)."""
    out = []
    for cfg in (2, 3, 5):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(cfg), "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-pmc", "--no-other-configs"]
        rec = {"baseline_config": CONFIGS[cfg]["key"], "steps": steps, "warmup": warmup}
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            r = d["roofline"]
            rec.update({"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "whole_chain_frac": r["whole_chain_frac"],
                        "kernel_frac": r["frac"], "kernel_avg_launch_ms": r["avg_launch_ms"], "receivers": d["config"]["receivers_per_gpu"],
                        "model": d["config"]["model"], "sample_rate": d["config"]["sample_rate"],
                        "algorithmic_bytes_per_sample": d["config"]["algorithmic_bytes_per_sample"],
                        "parity_checked": d["parity_checked"], "parity": d["parity"], "exit_status": p.returncode})
            for k in ("latency_ms_per_block", "realtime_factor"):
                if k in d:
                    rec[k] = d[k]
        except Exception as e:  # the main line must not depend on it
            rec["error"] = "%s: %s" % (type(e).__name__, e)
        out.append(rec)
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pin_to_gpu_numa_node(local_rank):
    """Bind this rank to the CPUs of its GPU's NUMA node; returns the node or None when the topology is not
    exposed (/sys/class/drm/card*/device/numa_node: the render nodes in PCI order are the HIP devices in the usual order)."""
    try:
        import glob
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() == "0x1002":   # AMD
                    cards.append((os.path.realpath(d), int(open(os.path.join(d, "numa_node")).read())))
            except Exception:
                pass
        cards.sort()
        if local_rank >= len(cards) or cards[local_rank][1] < 0:
            return None
        node = cards[local_rank][1]
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) and relay rank 0's line."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0].decode()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS), help="BASELINE.json config (SURVEY numbering): 4 = configs[3], the "
                    "metric's own (256 receivers x 1536 kSPS); 2 = configs[1] (one receiver, latency); 3 = configs[2] (6 MSPS, ModelChallenger)")
    ap.add_argument("--receivers", type=int, default=0, help="receivers per GPU (0 = the config's own: 256, or 1 for --config 2)")
    ap.add_argument("--preroll", type=int, default=40, help="untimed steps before the warm-up (GPU clock ramp)")
    ap.add_argument("--gpu-decode", action="store_true", help="also run the AIS::Decoder state machines on the device (frames out)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--parity-receivers", type=int, default=16, help="receivers compared with the oracle after the timed region (0 = off)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two short rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--tiles-per-span", type=int, default=0, help="front-end time tiling (0 = the library's choice); experiments")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise the rank launch / sharding / barrier / report path only (CPU tests)")
    ap.add_argument("--no-other-configs", action="store_true", help="do not time BASELINE configs[1] / configs[2] behind the default workload (other_configs in the line)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); they must agree" % (args.gpus, world))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")   # timing barrier + max over ranks only: no data-path collective, no RCCL

    import _pkg
    _pkg.load()
    from ais_catcher_amd import shard, workload

    C = CONFIGS[args.config]
    R = args.receivers if args.receivers > 0 else C["receivers"]
    rate, model, algo_bytes = C["rate"], C["model"], C["bytes"]
    nb = 2  # distinct resident blocks cycled (2 x 1.6 GB)
    rx_ids = shard.receiver_range(rank, world, R)  # this rank's receivers; no other rank touches them
    samples_per_step = R * BLOCK

    def barrier():
        if not args.dry_run:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    if args.dry_run:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.001 * args.steps)
        barrier()
        dt_local = time.perf_counter() - t0
        dt = shard.max_over_ranks(dt_local, dist)
        mine = [rx_ids[0], rx_ids[-1], round(dt_local / args.steps * 1e3, 4)]
        owned = [None] * world
        if dist is not None:
            dist.all_gather_object(owned, mine)
        else:
            owned = [mine]
        if rank == 0:
            print(json.dumps({"metric": "IQ Msamples/s (CFLOAT32) through ModelDefault chain", "dry_run": True,
                              "value": round(shard.aggregate_msamples(samples_per_step, world, args.steps, dt), 1), "unit": "Msamples/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                              "receiver_ranges": [o[:2] for o in owned],
                              "per_rank": [{"rank": i, "ms_per_step": o[2]} for i, o in enumerate(owned)]}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    from ais_catcher_amd import gpu
    torch.cuda.set_device(local)
    numa = pin_to_gpu_numa_node(local) if world > 1 else None   # one process per GPU on the GPU's NUMA node
    data = workload.resident_batch(torch, R, nb, seed=rx_ids[0] // R, unique=min(8, R), sample_rate=rate)
    fmt = C.get("format", "cf32")
    blocks_of = None
    if fmt == "cu8":   # Util::Convert's input: unsigned bytes, I then Q (the reference's file format `-r cu8`)
        data = torch.clamp(torch.round(data * 128.0 + 128.0), 0, 255).to(torch.uint8)
        torch.cuda.synchronize()
        blocks_of = lambda r: [data[b, r].cpu().numpy().reshape(-1) for b in range(nb)]
    in_fmt = {"cf32": gpu.FMT_CF32, "cu8": gpu.FMT_CU8}[fmt]
    g = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=BLOCK, device_id=local, gpu_decode=args.gpu_decode, model=model, tiles_per_span=args.tiles_per_span,
                   input_format=in_fmt)
    sequence = workload.block_sequence(args.preroll, args.warmup, args.steps, nb)
    it = iter(sequence)

    def step():
        g.submit_device(data[next(it)].data_ptr(), BLOCK)
        g.run()

    # Clock ramp: after the idle time of the set-up the GPU runs its first ~8 ms of load at a low clock and then pauses
    # for ~6 ms while it switches up (tools/host_times.py: run() call 5 of a cold context blocks for 7.5 ms).  A few
    # dozen untimed steps in front of the W warm-up steps keep that one-off transient out of the K timed steps.
    for _ in range(args.preroll):
        step()
    g.sync()
    for _ in range(args.warmup):
        step()
    g.sync()
    g.timing(not os.environ.get("BENCH_NO_K1_EVENTS"))

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0   # host time to enqueue all steps (includes the library's back-pressure: it lets the host run two blocks ahead)
    g.sync()
    barrier()
    dt_local = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt_local, dist)

    k1_ms, k1_n = g.frontend_ms()

    # ---- parity gate on the run that was just timed: the last block's outputs of a spread of receivers against the oracle
    n_checked, mismatches = 0, []
    frames_checked = None
    if args.parity_receivers > 0:
        frames = None
        try:
            g.sync_outputs()
        except gpu.AisGpuError as e:
            # --gpu-decode: a throughput run never collects its frames, so their ring has wrapped (AISGPU_ERR_OVERFLOW); the
            # decisions have been copied all the same
            if not (args.gpu_decode and "(5)" in str(e)):
                raise
        if args.gpu_decode:
            # ... so the gate looks at ONE MORE block, untimed, behind the timed ones: same decoders, same state, and this block's
            # frames alone are in the ring (decisions and frames of that block are compared)
            sequence = sequence + [(sequence[-1] + 1) % nb]
            g.submit_device(data[sequence[-1]].data_ptr(), BLOCK)
            g.run()
            g.sync_outputs()
            frames = g.frames()
            frames_checked = len(frames)
        n = min(args.parity_receivers, R)
        receivers = sorted(set(int(round(i * (R - 1) / max(n - 1, 1))) for i in range(n)))
        n_checked, mismatches = parity_check(g, data, sequence, receivers, rate=rate, model=model, frames=frames, fmt=fmt, blocks_of=blocks_of)

    # ---- host cost of one aisgpu_run() with the device idle (no back-pressure): what the calling thread pays per block
    host_ms = []
    for i in range(12):
        g.submit_device(data[i % nb].data_ptr(), BLOCK)
        t1 = time.perf_counter()
        g.run()
        host_ms.append((time.perf_counter() - t1) * 1e3)
        g.sync()
    host_ms = sorted(host_ms[2:])
    g.close()
    # the same kernel measured without the other streams' kernels competing for the chip (untimed extra steps)
    gs = gpu.AisGpu(sample_rate=rate, n_receivers=R, block_len=BLOCK, device_id=local, serial=True, model=model, input_format=in_fmt)
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
    value = shard.aggregate_msamples(samples_per_step, world, args.steps, dt)
    achieved = samples_per_step * algo_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    # HBM-side bytes per launch of the same kernel come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs:
    # tools/pmc_traffic.sh).  `traffic` is filled only from a measurement of THIS session (BENCH_TRAFFIC_JSON = the file that
    # script just wrote on this box); otherwise it is null and `traffic_reference` names the committed profile.
    traffic = traffic_bytes = traffic_src = step_bytes = None
    tpath = os.environ.get("BENCH_TRAFFIC_JSON")
    if tpath and os.path.exists(tpath) and R == 256 and args.config == 4 and k1_ms > 0:
        tj = json.load(open(tpath))
        traffic_bytes = tj["hbm_bytes_per_launch"]
        traffic = round(traffic_bytes / (k1_ms * 1e-3) / 1e9, 1)
        traffic_src = {"file": os.path.basename(tpath), "host": tj.get("host"), "commit": tj.get("commit")}
    elif world == 1 and rank == 0 and not args.no_pmc and k1_ms > 0 and not args.gpu_decode:
        tj = pmc_traffic_pass(args.config, R)   # (after the timed region and the contexts of this process are closed)
        if tj:
            traffic_bytes = tj["hbm_bytes_per_launch"]
            traffic = round(traffic_bytes / (k1_ms * 1e-3) / 1e9, 1)
            traffic_src = tj
            step_bytes = tj.get("step_bytes")
    ref_profile = None
    for cand in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if args.config == 4 and cand.startswith("r0") and cand.endswith("pmc_traffic_k1.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
                ref_profile = {"file": "profiles/" + cand, "hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
                               "host": tj.get("host"), "commit": tj.get("commit"), "note": "separate rocprofv3 --pmc session, not this run"}
            except Exception:
                pass
            break
    res = {
        "metric": "IQ Msamples/s (%s) through %s chain" % ("CFLOAT32" if fmt == "cf32" else fmt.upper(), C["model_name"]), "value": round(value, 1),
        "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
        "host_cost_ms_per_step": round(float(np.median(host_ms)), 4),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "parity_checked": n_checked, "parity": (("bit-exact vs oracle (hard bits, levels, ppm and the NMEA text of the device decoders' frames, block behind the last timed one)" if frames_checked is not None else "bit-exact vs oracle (hard bits, levels, ppm of the last timed block)") if n_checked and not mismatches
                                                else "off" if not n_checked else "MISMATCH: " + "; ".join(mismatches[:8])),
        "frames_checked": frames_checked,
        "config": {"workload": "BASELINE %s: %s, %d IQ samples per receiver per step, resident in HBM, chain up to hard bits/levels/ppm%s"
                               % (C["key"], (C["what"] % R) if "%d" in C["what"] else C["what"], BLOCK, " + FM-branch signs" if model == 4 else ""),
                   "baseline_config": C["key"], "model": C["model_name"], "algorithmic_bytes_per_sample": algo_bytes,
                   "gpu_frame_decoder": bool(args.gpu_decode), "receivers_per_gpu": R, "block_len": BLOCK, "sample_rate": rate, "parallelism": "receivers sharded, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_over_algorithmic": round(traffic_bytes / (samples_per_step * algo_bytes), 3) if traffic_bytes else None,
                     "traffic_bytes_per_launch": traffic_bytes, "traffic_source": traffic_src, "traffic_reference": ref_profile,
                     # all kernels of a step (front end, phasor recurrence, derotation / FIR, PhaseSearch, assemble), same PMC passes
                     "step_traffic_bytes": step_bytes,
                     "step_traffic_over_algorithmic": round(step_bytes / (samples_per_step * algo_bytes), 3) if step_bytes else None,
                     "algorithmic_bytes_per_launch": samples_per_step * algo_bytes,
                     "kernel": C["kernel"], "avg_launch_ms": round(k1_ms, 4), "launches": k1_n,
                     # the same algorithmic bytes over the whole step (all kernels of the chain, wall clock / steps)
                     "whole_chain_frac": round(samples_per_step * algo_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "isolated_launch_ms": round(iso_ms, 4),
                     "isolated_frac": round(samples_per_step * algo_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if iso_ms > 0 else None},
    }
    if args.config == 2:   # the single-receiver case: what a block costs end to end on the device, and against real time
        res["latency_ms_per_block"] = res["ms_per_step"]
        res["realtime_factor"] = round(BLOCK / rate / (dt / args.steps), 1)
    bad = len(mismatches)
    if dist is not None:  # a mismatch on any rank fails the job
        t = torch.tensor([bad], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        bad_all = int(t.item())
        c = torch.tensor([n_checked], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        res["parity_checked"] = int(c.item())
        if bad_all and not bad:
            res["parity"] = "MISMATCH on another rank"
        bad = bad_all
    if dist is not None:
        # every rank's own front-end launch time and wall clock (the line's roofline is rank 0's): min / max over the ranks
        per = [None] * world
        dist.all_gather_object(per, {"rank": rank, "k1_ms": round(k1_ms, 4), "ms_per_step": round(dt_local / args.steps * 1e3, 4), "numa_node": numa})
        res["per_rank"] = per
        res["roofline"]["avg_launch_ms_min_max_over_ranks"] = [min(p_["k1_ms"] for p_ in per), max(p_["k1_ms"] for p_ in per)]
    if rank == 0:
        if world == 1 and args.config == 4 and not args.no_other_configs and not args.no_pmc and not args.gpu_decode:  # (--no-pmc = the quick form of the tools' A/B loops)
            res["other_configs"] = other_configs()
            if any(str(o.get("parity", "")).startswith("MISMATCH") or o.get("exit_status") == 3 for o in res["other_configs"]):
                bad = max(bad, 1)   # parity of configs[1] / configs[2] is part of this run's gate
                res["parity"] = res["parity"] + "; MISMATCH in other_configs"
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, model=model, rate=rate)
        elif world > 1:
            res["cpu_baseline"] = "skipped (N>1: measured on rank 0 of the N=1 run only)"
            res["roofline"]["traffic_note"] = "PMC traffic is a single-GPU measurement (N=1 run)"
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 3 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
