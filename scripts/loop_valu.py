#!/usr/bin/env python3
"""static instruction mix of the ROW LOOP of a k_walk instantiation (hipcc -S, no GPU):
   python3 scripts/loop_valu.py [name-regex]   (default: the in-place pow2 walk)
The loop body holds three row steps (ring slots are compile-time), so per step = the numbers / 3."""
import re, sys, subprocess, collections, os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'point-cloud-preprocessing-tools_amd')
pat = sys.argv[1] if len(sys.argv) > 1 else r'k_walkILi2ELb1ELb0'
out = '/tmp/bev_loop.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
                       '-fhip-fp32-correctly-rounded-divide-sqrt', '-gline-tables-only', '-S', '--cuda-device-only', '-o', out,
                       'csrc/bev_kernels.hip'] + sys.argv[2:], cwd=root, stderr=subprocess.DEVNULL)
txt = open(out).read()
for m in re.finditer(r"^(_ZN4bevk\S+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.S | re.M):
    if not re.search(pat, m.group(1)):
        continue
    lines = m.group(2).split('\n')
    isins = lambda l: re.match(r'\s+[a-z][a-z0-9_]+(\s|$)', l) and not l.strip().startswith('.')
    total = sum(1 for l in lines if isins(l))
    lab = {}
    for i, l in enumerate(lines):
        lm = re.match(r'(\.LBB\d+_\d+):', l)
        if lm:
            lab[lm.group(1)] = i
    cands = {}
    for i, l in enumerate(lines):
        bm = re.match(r'\s+(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
        if bm and bm.group(2) in lab and lab[bm.group(2)] < i:
            a = lab[bm.group(2)]
            cands[a] = max(cands.get(a, 0), i)
    loops = sorted(((sum(1 for x in lines[a:i] if isins(x)), a, i) for a, i in cands.items()), reverse=True)
    print(m.group(1)[9:40], 'function instr', total, 'largest loops:', [(n, a, i) for n, a, i in loops[:5]])
    best = [l for l in reversed(loops) if sum(1 for x in lines[l[1]:l[2]] if 's_barrier' in x) >= 3][0] # the row loop: three steps, a barrier each
    n, a, b = best
    c = collections.Counter()
    for l in lines[a:b]:
        im = re.match(r'\s+([a-z][a-z0-9_]+)', l)
        if im and isins(l):
            c[im.group(1)] += 1
    g = lambda p: sum(v for k, v in c.items() if re.match(p, k))
    print(f"{m.group(1)[9:40]}: loop instr {n} valu {g('v_')} (readlane {g('v_readlane')} mov {g('v_mov')} cndmask {g('v_cndmask')}) "
          f"salu {g('s_')} (nop {g('s_nop')}) lds {g('ds_')} vmem {g('global_|buffer_')}")
