#!/bin/bash
# kernel resource notes of one object of vdn_hip/_build (dev): tools/dev/kres.sh sdf_lw_bf16 [asm-out]
O=/root/repo/vdn-nerf_amd/vdn_hip/_build/$1.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $O >/dev/null 2>&1
F=$O.0.hipv4-amdgcn-amd-amdhsa--gfx950
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $F 2>/dev/null | grep -E "\.name:|vgpr_count|sgpr_count|spill|private_segment_fixed|agpr" | paste - - - - - - - | sed 's/  */ /g' | cut -c1-260
[ -n "$2" ] && /opt/rocm/lib/llvm/bin/llvm-objdump -d $F > $2
rm -f $O.0.*
