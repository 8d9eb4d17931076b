#!/bin/bash
# Per-kernel register / scratch / LDS usage of one csrc/*.hip file as hipcc reports it for gfx950 (no GPU needed):
#   tools/kernel_resources.sh occdepth_amd/csrc/gemm_x3.hip [filter]
f=$(readlink -f "$1"); pat=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -c "$f" -Rpass-analysis=kernel-resource-usage -o /tmp/_kr.o 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' | grep -E "$pat" | while read -r line; do
   n=$(echo "$line" | sed -e 's/Function Name: \([^ ]*\).*/\1/'); d=$(echo "$n" | c++filt); echo "${line/$n/$d}"; done
