#!/bin/bash
# exp_libs/<name>.so = the product library with kpn_api.hip (every kernel but the pair-tile rows kernels) built with extra flags.
# Usage: build_api_variant.sh <name> [flags]
cd "$(dirname "$0")/.."; mkdir -p exp_libs /tmp/pv; name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize"
pair=keypointnerf_amd/_lib/geo_rows_pair_tu.o
[ -f $pair ] || python -m keypointnerf_amd.build
/opt/rocm/bin/hipcc $F "$@" -c keypointnerf_amd/csrc/kpn_api.hip -o /tmp/pv/api_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pv/api_$name.o $pair -o exp_libs/$name.so && echo built exp_libs/$name.so
