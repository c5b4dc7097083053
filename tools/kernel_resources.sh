#!/bin/bash
# VGPRs / SGPRs / scratch / occupancy of every kernel of the library, from hipcc's resource-usage remarks (no GPU needed):
#   tools/kernel_resources.sh [file.hip ...]        (default: all kernel files)
cd "$(dirname "$0")/../clfacedetection_amd/csrc" || exit 1
files=${@:-vj_kernels.hip vj_cv_profile.hip vj_cv_tile.hip vj_group_dev.hip}
for f in $files; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -DVJ_BUILDING \
        -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
        grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - |
        sed 's/Function Name: //' | while IFS=$'\t' read -r name rest; do printf '%s | %s\n' "$(echo "$name" | c++filt | cut -c1-100)" "$(echo "$rest" | tr '\t' ' ')"; done
done
