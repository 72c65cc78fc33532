#!/bin/bash
# registers / spills / occupancy of every kernel of one csrc/*.hip file (compiles to a scratch object)
# usage: tools/kernel_resources.sh decoder16 [extra hipcc flags]
f=$1; shift
cd "$(dirname "$0")/../pangnn_amd/csrc" || exit 1
flags=$(make -s -p -n 2>/dev/null | grep "^FLAGS_$f" | sed 's/.*= *//')
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags "$@" -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2>&1 |
  grep -E "error|Function Name|VGPRs:|VGPRs Spill|Occupancy|LDS Size" |
  sed 's/^[^ ]* remark: *//; s/\[-Rpass-analysis=kernel-resource-usage\]//; s/Function Name: /\n/' | paste -sd' ' | sed 's/ *\(_Z\)/\n\1/g' | sed 's/\(_Z[A-Za-z0-9_]\{50\}\)[A-Za-z0-9_]*/\1/'
echo
