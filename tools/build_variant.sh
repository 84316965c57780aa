#!/bin/bash
# build/variants/libhns_<name>.so with extra -D flags: tools/build_variant.sh <name> [-DTP_PK=1 ...]
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p build/variants
S=multi-uav-pursuit-evasion_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-unused-value -Iinclude "$@" \
  $S/hns_kernels.hip $S/hns_tp.hip $S/hns_envgen.hip -o build/variants/libhns_$N.so
echo build/variants/libhns_$N.so
