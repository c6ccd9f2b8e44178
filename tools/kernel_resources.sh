#!/bin/bash
# registers / spills / LDS / occupancy of every kernel in one .hip file (hipcc -Rpass-analysis)
#   tools/kernel_resources.sh svinet_amd/csrc/svils_lpl.hip          (EXTRA="-DFLAG=1 ..." adds compile flags, e.g. an A/B macro)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage $EXTRA 2>&1 |
  grep "remark:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' |
  awk '/^Function Name/ { if (n) print n, r; n=$3; r=""; next }
       /^(VGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size|TotalSGPRs)/ { r = r " | " $0 }
       END { print n, r }' | while read -r name rest; do echo "$(echo "$name" | c++filt | cut -c1-60) $rest"; done
