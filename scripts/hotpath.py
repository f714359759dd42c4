"""Instruction tally of chosen line ranges of an AMDGPU assembly file (the common path of a loop, picked by hand):
python3 scripts/hotpath.py file.s 1456-1469 1470-1507 ...   prints vector / scalar / LDS / memory counts per range and in all,
and the vector pipe's weighted cycles from the calibrated costs (profiles/r02_roofcal_ops.txt: add/sub/mov/and/or/xor/
lshr/ashr 2.4 cycles, everything else 4.25; s_nop N counts as one scalar instruction)."""
import sys, re
L = open(sys.argv[1]).read().split('\n')
cheap = re.compile(r'^v_(add_u32|sub_u32|subrev_u32|mov_b32|and_b32|or_b32|xor_b32|lshrrev_b32|ashrrev_i32|not_b32)(_e32|_e64)?$')
tot = dict(v=0, s=0, lds=0, mem=0, w=0.0, nop=0)
for r in sys.argv[2:]:
    a, b = map(int, r.split('-'))
    c = dict(v=0, s=0, lds=0, mem=0, w=0.0, nop=0)
    for l in L[a - 1:b]:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith('v_'):
            c['v'] += 1
            c['w'] += 2.4 if (cheap.match(op) and 'dpp' not in t and 'sdwa' not in t) else 4.25
        elif op.startswith('s_'):
            c['s'] += 1
            if op == 's_nop': c['nop'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('global_', 'flat_', 'scratch_', 'buffer_')): c['mem'] += 1
    print("%-12s v=%-3d s=%-3d (nop %d) lds=%-2d mem=%-2d weighted %.0f" % (r, c['v'], c['s'], c['nop'], c['lds'], c['mem'], c['w']))
    for k in tot: tot[k] += c[k]
print("%-12s v=%-3d s=%-3d (nop %d) lds=%-2d mem=%-2d weighted %.0f" % ("all", tot['v'], tot['s'], tot['nop'], tot['lds'], tot['mem'], tot['w']))
