#!/bin/bash
# Code-placement variants of the library for the mode-2 soak (exp_libs/soak_*.so): the whole k_geo_rows_h2 body shifted by
# 4 * PAD bytes (s_nop sled at kernel entry), and three different scheduling-region granularities (different instruction
# orders altogether).  Usage: bash scripts/build_soak_variants.sh
cd "$(dirname "$0")/.."; mkdir -p exp_libs
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize"
S=keypointnerf_amd/csrc/kpn_api.hip
n=0
for spec in "PAD=1" "PAD=3" "PAD=7" "PAD=13" "PAD=16" "PAD=29" "PAD=64" "PAD=129" "PAD=5 -DKPN_H2_REGIONS=4" "PAD=11 -DKPN_H2_REGIONS=2" "PAD=2 -DKPN_H2_REGIONS=1"; do
  /opt/rocm/bin/hipcc $F -DKPN_H2_$spec $S -o exp_libs/soak_$n.so &
  n=$((n+1)); if (( n % 4 == 0 )); then wait; fi
done
wait; ls exp_libs/soak_*.so | wc -l
