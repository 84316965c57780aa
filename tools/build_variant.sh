#!/bin/bash
# build/variants/libhns_<name>.so with extra -D flags: tools/build_variant.sh <name> [-DTP_PK=1 ...]   (load it with HNS_LIBRARY=<path>)
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p build/variants
HNS_BUILD_OUT=build/variants/libhns_$N.so HNS_EXTRA_FLAGS="$*" python -c "import __graft_entry__ as g; g.build(force=True)"
echo build/variants/libhns_$N.so
