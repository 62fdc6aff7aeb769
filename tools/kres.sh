#!/bin/bash
# usage: tools/kres.sh <file.hip> [name-filter]   — compact per-kernel resource table (VGPRs, spills, LDS, occupancy)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Ivkfft_amd/csrc -Wno-unused-result --offload-arch=gfx950 -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | awk '/Function Name:/{name=$(NF-1)} /TotalSGPRs:/{s=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /Occupancy/{o=$(NF-1)} /LDS Size/{l=$(NF-1); print name, "sgpr="s, "vgpr="v, "agpr="a, "scratch="sc, "occ="o, "lds="l}' \
 | grep -E "${2:-.}" | while read n rest; do echo "$(echo $n | c++filt | sed 's/vkfft_mi355x:://g; s/void //; s/(.*//' | cut -c1-120) $rest"; done
