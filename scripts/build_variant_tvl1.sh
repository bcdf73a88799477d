#!/bin/bash
# Build a tuning variant of the library with flags for the TV-L1 kernels: scripts/build_variant_tvl1.sh <name> <extra hipcc flags...>
# -> variants/libsrx_<name>.so (select it with SRX_LIBRARY=<path>)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/build/var_$name
mkdir -p "$obj" "$root/variants"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$root/include"
cd "$root/superresolution_amd/csrc"
/opt/rocm/bin/hipcc "$@" $F -c srx_tvl1.hip -o "$obj/srx_tvl1.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/variants/libsrx_$name.so" "$root/build/srx/srx_kernels.o" \
  "$obj/srx_tvl1.o" "$root/build/srx/srx_farneback.o" "$root/build/srx/srx_quality.o" "$root/build/srx/srx_solver.o" "$root/build/srx/srx_api.o"
echo "variants/libsrx_$name.so"
