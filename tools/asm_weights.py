#!/usr/bin/env python3
"""tools/asm_weights.py <kernel.s> : vector-issue cycles per phase of a chain kernel, from its assembly (built with -DDH_ASM_MARKERS,
tools/asm_census.py) and the per-instruction issue costs measured on an MI355X (tools/microbench/valu_rate.hip, >= 2 wavefronts per
SIMD): 2.4 cycles for plain f32 add / sub / mul / fma and 32-bit integer add / logic / move, 4.2 for everything else on the vector
ALU (packed f32, conversions, min / max, shifts, compares, selects, DPP, readlane), 8.2 for v_fma_mix*, 16 for v_mfma 16x16x32.
Counts the instructions in TEXT order between the phase markers of the run loop (rare paths are laid out behind the loop, so the
text between two markers is mostly the hot path; blocks that only a not-taken branch reaches are still counted: an upper bound)."""
import re
import sys
from collections import Counter, defaultdict

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_co_u32", "v_not_b32"}


def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_mfma"):
        return 16.0
    if base.startswith("v_fma_mix"):
        return 8.2
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return 4.2
    return 2.4 if base in FAST else 4.2


def main():
    phase, per, cls = "pre", defaultdict(float), defaultdict(Counter)
    order = []
    for line in open(sys.argv[1]):
        m = re.search(r"; DH_PHASE (\w+)", line)
        if m:
            phase = "after " + m.group(1)
            if phase not in order:
                order.append(phase)
            continue
        t = line.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_"):
            per[phase] += cost(op)
            cls[phase][re.sub(r"_(e32|e64)$", "", op)] += 1
        elif op.startswith(("ds_", "global_", "scratch_", "buffer_", "s_")):
            cls[phase]["[" + op.split("_")[0] + "]"] += 1
    tot = 0.0
    for ph in order:
        n = sum(c for k, c in cls[ph].items() if k.startswith("v_"))
        tot += per[ph]
        top = ", ".join("%s %d" % kv for kv in cls[ph].most_common(9))
        print("%-12s %5d vector instr %8.0f cycles | %s" % (ph, n, per[ph], top))
    print("total (text order, incl. rare blocks inside the loop) %.0f cycles" % tot)


if __name__ == "__main__":
    main()
