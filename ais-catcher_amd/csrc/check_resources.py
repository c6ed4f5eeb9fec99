#!/usr/bin/env python3
"""Build-time check of the register budgets the hot kernels are written for (called by the Makefile with the compiler's
-Rpass-analysis=kernel-resource-usage remarks).  `amdgpu_num_vgpr(N / 2)` on k4_phase_chunks / k4_box_chunks relies on how this
LLVM counts the unified register file of gfx90a+ targets (the attribute gets half of the registers the kernel may use); another
compiler that takes the number literally would give PhaseSearch 32 / 48 registers and spill its state to scratch -- results stay
exact, the step gets much slower.  A budget that is not met fails the build instead of shipping that silently."""
import re
import sys

# kernel-name fragment -> (most VGPRs, most scratch bytes per lane).  VGPR ceilings = the residency the stream plan assumes
# (DESIGN.md section 5): three front-end waves per SIMD (136), one PhaseSearch wave beside them (64 / 96).
BUDGET = [
    ("k4_phase_chunks", 64, 0),
    ("k4_box_chunks", 96, 64),          # (the boxcar ring: 44 bytes of scratch for rarely used state, measured with them)
    ("k1_dppILi4E", 136, 0),            # every format of the four-stage front end, with and without the pre-decimation output
    ("k1_dppILi5E", 192, 0),
    ("k1_dppILi6E", 256, 0),
    ("k6_window_fir", 64, 0),
    ("k2_cgf_phasor_ck_pairs", 128, 0),
    ("k1u_resample_frontend", 128, 0),
    ("kv2_engine_rolesE", 256, 0),       # three waves per channel, as it compiles (two waves per SIMD)
    ("kv2_engine_roles_dense", 168, 128), # the same under 168 registers (three waves per SIMD; measured with its 112 bytes of scratch)
    ("kv2_engineE", 256, 0),           # (kv2_engine: one wave per channel; kv2_engine_roles: three)
    ("k1x_wave", 168, 0),              # one-wave front ends of round 6 (mode X at 96 kSPS / the decimate-by-3 tail / dual-channel 96 kSPS): four resp. two waves per SIMD, no scratch
    ("k1k_wave", 256, 0),
    ("kv2_estimate", 128, 0),
]


def main(path):
    text = open(path, errors="replace").read()
    recs = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", text, flags=re.S)
    if not recs:
        print("check_resources: no kernel-resource-usage remarks in %s" % path, file=sys.stderr)
        return 1
    bad = 0
    for frag, vmax, smax in BUDGET:
        hits = [(n, int(v), int(s)) for n, v, s in recs if frag in n]
        if not hits:
            print("check_resources: no kernel matches %r" % frag, file=sys.stderr)
            bad += 1
        for n, v, s in hits:
            if v > vmax or s > smax:
                print("check_resources: %s uses %d VGPRs / %d bytes of scratch, budget %d / %d" % (n, v, s, vmax, smax), file=sys.stderr)
                bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
