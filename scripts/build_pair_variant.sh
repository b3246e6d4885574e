#!/bin/bash
# exp_libs/<name>.so = the product library with the pair-kernel TU built with extra flags.  Usage: build_pair_variant.sh <name> [flags]
# Flags that change the packed streams as well (e.g. -DKPN_H2_LOG2ACT=0) must reach both translation units: API_FLAGS="..." too.
cd "$(dirname "$0")/.."; mkdir -p exp_libs /tmp/pv; name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize"
api=/tmp/pv/kpn_api.o
if [ -n "$API_FLAGS" ]; then
  api=/tmp/pv/kpn_api_$name.o
  /opt/rocm/bin/hipcc $F $API_FLAGS -c keypointnerf_amd/csrc/kpn_api.hip -o $api || exit 1
else
  fresh=1
  for f in keypointnerf_amd/csrc/*.hip keypointnerf_amd/csrc/*.h include/kpnerf.h; do [ -f $api ] && [ $api -nt $f ] || fresh=0; done
  [ $fresh == 1 ] || /opt/rocm/bin/hipcc $F -c keypointnerf_amd/csrc/kpn_api.hip -o $api || exit 1
fi
/opt/rocm/bin/hipcc $F "$@" -c keypointnerf_amd/csrc/geo_rows_pair_tu.hip -o /tmp/pv/pair_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $api /tmp/pv/pair_$name.o -o exp_libs/$name.so && echo built exp_libs/$name.so
