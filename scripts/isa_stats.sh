# usage: bash scripts/isa_stats.sh [kernel-name-regex]  -> registers, spills, LDS and instruction mix of the device kernels
# (static counts from hipcc -S; no GPU needed)
cd "$(dirname "$0")/../point-cloud-preprocessing-tools_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -S --cuda-device-only -o /tmp/bev_k.s csrc/bev_kernels.hip 2>/dev/null || exit 1
python3 - "${1:-.}" <<'PY'
import re,sys
pat=re.compile(sys.argv[1])
txt=open('/tmp/bev_k.s').read()
meta={}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    d=dict(re.findall(r'\.(\w+):\s+(\d+)', m.group(2))); meta[m.group(1)]=d
for m in re.finditer(r"^(_ZN4bevk\S+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.S|re.M):
    name=m.group(1)
    if not pat.search(name): continue
    body=m.group(2); ins=re.findall(r'^\s+([a-z][a-z0-9_]+)', body, re.M)
    c=lambda p: sum(1 for i in ins if re.match(p,i))
    d=meta.get(name,{})
    print(f"{name[9:60]:52s} vgpr {d.get('vgpr_count','?'):>3} sgpr_spill {d.get('sgpr_spill_count','?'):>3} lds {d.get('group_segment_fixed_size','?'):>6} | instr {len(ins):5d} valu {c('v_'):5d} (readlane {c('v_readlane'):4d} writelane {c('v_writelane'):3d}) salu {c('s_'):5d} lds {c('ds_'):4d} vmem {c('global_|buffer_'):3d} waitcnt {c('s_waitcnt'):3d}")
PY
