#!/bin/bash
# Print VGPR/AGPR/occupancy/spill per kernel of one .hip file:  tools/kernel_resources.sh conv.hip
cd "$(dirname "$0")/../pb_sed_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c "$1" -o /tmp/_kr.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|VGPRs Spill|ScratchSize|Occupancy|LDS Size" | \
  sed -E 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - - - | \
  sed -E 's/Function Name: _ZN5pbsed[0-9]*//; s/EvNS_.*E\t/\t/' | cut -c1-200
