#!/bin/bash
# exp_libs/<name>.so = the product library with the pair-kernel TU built with extra flags.  Usage: build_pair_variant.sh <name> [flags]
cd "$(dirname "$0")/.."; mkdir -p exp_libs /tmp/pv; name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
[ -f /tmp/pv/kpn_api.o ] && [ /tmp/pv/kpn_api.o -nt keypointnerf_amd/csrc/kpn_api.hip ] && [ /tmp/pv/kpn_api.o -nt keypointnerf_amd/csrc/field_kernels.hip ] || /opt/rocm/bin/hipcc $F -c keypointnerf_amd/csrc/kpn_api.hip -o /tmp/pv/kpn_api.o
/opt/rocm/bin/hipcc $F -fno-slp-vectorize "$@" -c keypointnerf_amd/csrc/geo_rows_pair_tu.hip -o /tmp/pv/pair_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pv/kpn_api.o /tmp/pv/pair_$name.o -o exp_libs/$name.so && echo built exp_libs/$name.so
