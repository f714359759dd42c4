"""SIMD time of the vector instructions of a stretch of AMDGPU assembly, weighted by what each KIND of instruction costs
(profiles/r02_roofcal_ops.txt: `damar_amd/bin/roofcal ops`, cycles per wave64 instruction at 5 wavefronts per SIMD): 2.5-3.0
cycles for v_add / v_sub / v_mov_b32 / v_and / v_or / v_xor / v_lshrrev_b32 / v_ashrrev_i32, 4.4 for everything else measured.

  python3 scripts/valu_weight.py pp0.s <first line> <last line>     (the wave loop of pk_pass<REV>, see scripts/asm_blocks.py)
"""
import collections
import sys

CHEAP = {"v_add_u32": 3.0, "v_sub_u32": 2.9, "v_subrev_u32": 2.9, "v_mov_b32": 2.5, "v_and_b32": 2.55, "v_or_b32": 2.55,
         "v_xor_b32": 2.5, "v_lshrrev_b32": 2.5, "v_ashrrev_i32": 2.5, "v_not_b32": 2.5}
OTHER = 4.4


def main():
    lines = open(sys.argv[1]).read().split("\n")[int(sys.argv[2]):int(sys.argv[3])]
    n = collections.Counter()
    for l in lines:
        t = l.strip()
        if not t.startswith("v_"):
            continue
        op = t.split()[0]
        for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
            if op.endswith(suf):
                op = op[:-len(suf)] + ("_dpp" if suf == "_dpp" else "")
        n[op] += 1
    tot = sum(n.values())
    cyc = sum(c * CHEAP.get(op, OTHER) for op, c in n.items())
    cheap = sum(c for op, c in n.items() if op in CHEAP)
    print("%d vector instructions, %d (%.0f %%) of the cheap kind; weighted %.2f cycles of SIMD time each" % (tot, cheap, 100. * cheap / tot, cyc / tot))
    for op, c in n.most_common(14):
        print("  %-22s %4d  x %.2f" % (op, c, CHEAP.get(op, OTHER)))


if __name__ == "__main__":
    main()
