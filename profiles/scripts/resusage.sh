#!/bin/bash
# resusage.sh <file.hip> [pattern] [extra hipcc flags...]: registers / spills / occupancy / LDS of the kernels of one source file
f=$1; pat=${2:-.}; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -I../../include -I. -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/resusage.o 2>&1 \
 | grep -E "Function Name|VGPRs:|VGPRs Spill|Occupancy|LDS Size|error|warning:" | sed -e 's/^[A-Za-z0-9_]*\.h[ip]*:[0-9:]* remark: *//' -e 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' \
 | awk '/Function Name/{if(line)print line; line=$3} /VGPRs:/{line=line" vgpr="$2} /Spill/{line=line" spill="$3} /Occupancy/{line=line" occ="$3} /LDS/{line=line" lds="$4} /error|warning:/{print} END{print line}' | c++filt | grep -E "$pat"
