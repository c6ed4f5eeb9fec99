#!/usr/bin/env python3
"""Per-kernel "issue ms / HBM ms / measured ms" of one bench step (profiles/rNN_issue_model.txt): the budget the judge asked for.

  issue ms   = SQ_INSTS_VALU (dynamic wave64 VALU instructions of the launch, rocprofv3 --pmc, serial mode)
               x the cost of the kernel's instruction mix (static mix of its hottest loop from the ISA listing, cost per class from
                 tools/microbench_issue.hip at 4 waves per SIMD) / (1024 SIMDs x clock)
  HBM ms     = FETCH_SIZE x 2 + WRITE_SIZE of the launch / 6.29 TB/s (what MI355X_MICROARCH.md calls achievable)
  measured   = the launch alone (serial mode) and next to the other kernels (the bench run), rocprofv3 --kernel-trace --stats
usage: tools/issue_model.py kernels.s pmc_sq.txt pmc_traffic_all.txt serial_stats.txt bench_stats.txt mb_issue.txt"""
import re
import subprocess
import sys
from collections import Counter

s_path, sq_path, tr_path, ser_path, ovl_path, mb_path = sys.argv[1:7]

# ---- class costs (SIMD cycles per wave64 instruction, W = 4 column) and clock
cost, clock = {}, 2.35
for line in open(mb_path):
    m = re.match(r"^(\S.*?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m:
        cost[m.group(1).strip()] = float(m.group(4))
    m = re.search(r"clock from the s_nop loop: ([\d.]+) GHz", line)
    if m:
        clock = float(m.group(1))
C_PLAIN = cost.get("v_add_f32", 2.5)
C_PK = cost.get("v_pk_add_f32", 4.23)
C_DPP = cost.get("v_mov_b32_dpp wave_shr:1", 4.35)
C_CMP = cost.get("v_cmp_gt_f32 -> sgpr pair", 4.59)
C_VOP3 = cost.get("v_add3_u32", 4.22)   # three-operand / 64-bit integer forms (v_bfe, v_add3, v_lshl*_b64, v_and_or, v_fma_f64 ...)
C_F64 = cost.get("v_fma_f64", 4.8)


def klass(m, ops):
    if not m.startswith("v_"):
        return None
    if m.startswith("v_pk_"):
        return "packed"
    if "dpp" in ops or m.endswith("_dpp"):
        return "dpp"
    if m.startswith("v_cmp"):
        return "cmp"
    if m.endswith("_f64") or m.endswith("_b64") or m.endswith("_u64") or m.endswith("_i64"):
        return "wide"
    if re.match(r"v_(bfe|bfi|add3|and_or|or3|lshl_or|lshl_add|xad|perm|alignbit|mad_|fma_|med3|min3|max3|cndmask|readlane|writelane|readfirstlane)", m) or m.endswith("_e64"):
        return "vop3"
    if m.startswith("v_"):
        return "plain"
    return None


COST = {"plain": C_PLAIN, "packed": C_PK, "dpp": C_DPP, "cmp": C_CMP, "vop3": C_VOP3, "wide": C_F64}


def hot_mix(kernel_key):
    """static VALU class mix of the kernel's longest loop (falls back to the whole kernel)"""
    lines = open(s_path).read().splitlines()
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kernel_key in l), None)
    if start is None:
        return None
    end = start + 1
    while end < len(lines) and not lines[end].startswith(".Lfunc_end"):
        end += 1
    labels, insts = {}, []
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)$", l)
        if m and not m.group(1).startswith("."):
            insts.append((m.group(1), m.group(2)))
    loops = []
    for i, (m, o) in enumerate(insts):
        if m.startswith("s_cbranch") or m == "s_branch":
            tgt = o.strip().split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    a, b = max(loops, key=lambda t: t[1] - t[0]) if loops else (0, len(insts) - 1)
    c = Counter(k for k in (klass(m, o) for m, o in insts[a:b + 1]) if k)
    return c


def counters(path):
    out = {}
    for l in open(path):
        m = re.match(r"^(.*?)\s{2,}(SQ_\w+)\s+([\d.]+)\s*$", l)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    return out


def stats(path):
    out = {}
    for l in open(path):
        m = re.match(r"^(.*?)\s{2,}(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", l)
        if m:
            out[m.group(1).strip()] = float(m.group(4))
    return out


def traffic(path):
    out = {}
    for l in open(path):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m:
            out[m.group(1).strip()] = float(m.group(3)) + float(m.group(4))
    return out


sq, ser, ovl, tr = counters(sq_path), stats(ser_path), stats(ovl_path), traffic(tr_path)
rows = [("front end  k1_dpp<4,0,false> (incl. spectral analysis)", "k1_dppILi4ELi0ELb0", "k1_dpp<4, 0, false>"),
        ("PhaseSearch  k4_phase_chunks", "k4_phase_chunks", "k4_phase_chunks"),
        ("derotation + FIR  k3_derot_fir", "k3_derot_firILi0", "k3_derot_fir<0>"),
        ("phasor recurrence  k2_cgf_phasor_ck (8 CUs of its own)", "k2_cgf_phasor_ckE", "k2_cgf_phasor_ck"),
        ("assemble  k4_assemble", "k4_assemble", "k4_assemble")]
print("issue-cost classes (cycles per wave64 instruction, 4 waves per SIMD, clock %.2f GHz): plain VOP1/2 f32 & logic %.2f, packed f32 %.2f, DPP %.2f, "
      "compare %.2f, three-operand / VOP3 integer %.2f, 64-bit & f64 %.2f" % (clock, C_PLAIN, C_PK, C_DPP, C_CMP, C_VOP3, C_F64))
print()
print("%-58s %9s %-44s %8s %9s %8s %8s %9s %9s" % ("kernel (one launch of a 256-receiver step)", "VALU M", "static mix of the hot loop (%)", "cyc/inst", "issue ms", "HBM MB", "HBM ms", "alone ms", "beside ms"))
tot_issue = tot_hbm = 0.0
for label, skey, name in rows:
    cs = next((v for k, v in sq.items() if name in k), {})
    n_valu = cs.get("SQ_INSTS_VALU", 0.0)
    mix = hot_mix(skey) or Counter()
    n = sum(mix.values()) or 1
    avg = sum(COST[k] * v for k, v in mix.items()) / n
    issue_ms = n_valu * avg / (1024 * clock * 1e9) * 1e3
    if "phasor" in name:
        issue_ms = n_valu * avg / (32 * clock * 1e9) * 1e3   # confined to 8 CUs = 32 SIMDs
    mb = next((v for k, v in tr.items() if name in k), 0.0)
    hbm_ms = mb * 1e6 / 6.29e12 * 1e3
    alone = next((v for k, v in ser.items() if name in k), 0.0) / 1e3
    beside = next((v for k, v in ovl.items() if name in k), 0.0) / 1e3
    mixs = " ".join("%s %d" % (k, round(100.0 * v / n)) for k, v in mix.most_common())
    print("%-58s %9.1f %-44s %8.2f %9.3f %8.0f %8.3f %9.3f %9.3f" % (label, n_valu / 1e6, mixs, avg, issue_ms, mb, hbm_ms, alone, beside))
    if "phasor" not in name:
        tot_issue += issue_ms
    tot_hbm += hbm_ms
print()
print("sum over the kernels that share the 248 CUs: VALU issue %.3f ms, HBM-side traffic at 6.29 TB/s %.3f ms per step" % (tot_issue, tot_hbm))
