#!/bin/bash
# Register / spill / LDS usage of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
cd "$(dirname "$0")/../keypointnerf_amd/csrc"
for tu in "kpn_api.hip" "geo_rows_pair_tu.hip -fno-slp-vectorize"; do
  set -- $tu
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "${@:2}" $EXTRA -Rpass-analysis=kernel-resource-usage -c $1 -o /dev/null 2>&1 |
  awk '/Function Name:/{name=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /SGPRs Spill:/{ss=$(NF-1)} /VGPRs Spill:/{vs=$(NF-1)} /ScratchSize/{sc=$(NF-2)} /Occupancy/{oc=$(NF-1)} /LDS Size/{printf "%-44s vgpr %3s agpr %3s sgpr_spill %3s vgpr_spill %3s scratch %4s occ %s lds %s\n", substr(name,1,44), v, a, ss, vs, sc, oc, $(NF-2)}'
done
