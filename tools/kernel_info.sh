#!/bin/bash
# registers, spills and LDS of every kernel in a built library: tools/kernel_info.sh [lib.so] [name filter]
LIB=${1:-ray_tracing_amd/lib/librt_hip.so}; PAT=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$LIB" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.o 2>/dev/null || \
  { /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB" && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.o; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.o | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s*(\S+)',blk) or [0,'?'])[1]
    name=g('name')
    if re.search(sys.argv[1],name): print('%-90s vgpr %s agpr %s sgpr %s spill_v %s lds %s scratch %s' % (name[:90], g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('vgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
" "$PAT"
rm -rf $T
