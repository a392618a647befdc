#!/bin/bash
# VGPR / spill / LDS numbers of every kernel in a hipcc object (or libnudf*.so): scripts/kernel_regs.sh <file> [name filter]
set -e
f=$1; pat=${2:-.}
t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$t/fb.bin "$f" 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$t/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/dev.o | python3 -c "
import sys,re
cur={}
rows=[]
for ln in sys.stdin:
    m=re.match(r'\s+\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)',ln)
    if not m: continue
    k,v=m.groups()
    if k in cur and k=='name' : pass
    cur[k]=v
    if len(cur)>=7 and 'name' in cur and 'vgpr_count' in cur and 'group_segment_fixed_size' in cur and 'vgpr_spill_count' in cur and 'private_segment_fixed_size' in cur and 'sgpr_count' in cur and 'sgpr_spill_count' in cur:
        rows.append(cur); cur={}
import subprocess
for r in rows:
    name=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    if re.search(r'''$pat''',name):
        print('%-70s vgpr %3s agpr %3s spill %3s scratch %5s lds %6s sgpr %3s'%(name[:70],r['vgpr_count'],r.get('agpr_count','-'),r['vgpr_spill_count'],r['private_segment_fixed_size'],r['group_segment_fixed_size'],r['sgpr_count']))
"
rm -rf $t
