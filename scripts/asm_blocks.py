"""Per-basic-block instruction counts of a stretch of AMDGPU assembly (hipcc -S --cuda-device-only): vector instructions
(how many of them register moves / v_readlane reloads of spilled SGPRs), scalar, LDS, memory, and the block's branches.
Used on the wave loop of pk_pass<REV> (kernels/report_packed.h) to see what a wave step executes:

  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Idamar_amd/csrc -S --cuda-device-only -o report.s damar_amd/csrc/kernels/report.hip
  awk '/^_Z7pk_passILi0E/,/\.Lfunc_end/' report.s > pp0.s        # one direction
  python3 scripts/asm_blocks.py pp0.s <first line> <last line>   # the loop: around the third global_load_dwordx2
"""
import sys,re
lines=open(sys.argv[1]).read().split('\n')
lo,hi=int(sys.argv[2]),int(sys.argv[3])
blocks=[];cur=None
for n,l in enumerate(lines[lo:hi],lo):
    t=l.strip()
    if not t or t.startswith(';'): continue
    m=re.match(r'^(\.LBB\d+_\d+):',t)
    if m:
        cur={'name':m.group(1),'line':n,'v':0,'s':0,'mov':0,'lds':0,'mem':0,'rl':0,'br':[], 'cmt':l.split(';',1)[1].strip() if ';' in l else ''}
        blocks.append(cur); continue
    if t.startswith('; %bb.'):
        cur={'name':t.split()[1].rstrip(':'),'line':n,'v':0,'s':0,'mov':0,'lds':0,'mem':0,'rl':0,'br':[],'cmt':''}
        blocks.append(cur); continue
    if cur is None: continue
    op=t.split()[0]
    if op.startswith('v_'):
        cur['v']+=1
        if op.startswith('v_mov'): cur['mov']+=1
        if 'readlane' in op: cur['rl']+=1
    elif op.startswith('s_'):
        cur['s']+=1
        if op.startswith('s_cbranch') or op=='s_branch': cur['br'].append(op.replace('s_cbranch_','')+'->'+t.split()[-1])
    elif op.startswith('ds_'): cur['lds']+=1
    elif op.startswith(('global_','flat_','scratch_','buffer_')): cur['mem']+=1
for b in blocks:
    print("%-12s L%-5d v=%-3d (mov %-2d rl %d) s=%-3d lds=%-2d mem=%-2d %s" % (b['name'],b['line'],b['v'],b['mov'],b['rl'],b['s'],b['lds'],b['mem'],' '.join(b['br'])))
