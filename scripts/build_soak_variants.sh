#!/bin/bash
# Code-placement variants of the library for the mode-2 soak (exp_libs/soak_*.so): the whole k_geo_rows_h2 body shifted by
# 4 * PAD bytes (s_nop sled at kernel entry); with and without the look-ahead into the next layer (different instruction
# streams altogether).  Usage: bash scripts/build_soak_variants.sh
cd "$(dirname "$0")/.."; mkdir -p exp_libs
n=0
for spec in "-DKPN_H2_PAD=1" "-DKPN_H2_PAD=3" "-DKPN_H2_PAD=7" "-DKPN_H2_PAD=13" "-DKPN_H2_PAD=16" "-DKPN_H2_PAD=29" "-DKPN_H2_PAD=64" "-DKPN_H2_PAD=129" \
            "-DKPN_H2_PAD=5 -DKPN_H2_LOOKAHEAD=false" "-DKPN_H2_PAD=11 -DKPN_H2_LOOKAHEAD=false" "-DKPN_H2_PAD=2 -DKPN_H2_LOOKAHEAD=false"; do
  bash scripts/build_pair_variant.sh soak_$n $spec > /dev/null 2>&1 &
  n=$((n+1)); if (( n % 4 == 0 )); then wait; fi
done
wait; ls exp_libs/soak_*.so | wc -l
