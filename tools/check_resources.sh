#!/bin/bash
# Register / scratch report of every kernel in neurosis_amd/csrc (cross-compiles without a GPU).  A kernel that shows scratch bytes or
# VGPR spills here has a run-time-indexed register array or exceeds its launch-bounds register cap: fix before measuring.
#   bash tools/check_resources.sh            # kernels with scratch or spills only
#   bash tools/check_resources.sh all        # every kernel: name, VGPRs, scratch bytes per lane, spilled VGPRs
cd "$(dirname "$0")/../neurosis_amd/csrc" || exit 1
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -c "$f" -o /tmp/_res_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
    | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill" | sed 's/.*remark: //;s/\[-Rpass.*//' | paste - - - - \
    | awk -v all="$1" -v file="$f" '{ if (all == "all" || $0 !~ /ScratchSize \[bytes\/lane\]: 0 / || $0 !~ /VGPRs Spill: 0/) print file ": " $0 }'
done
rm -f /tmp/_res_$$.o
