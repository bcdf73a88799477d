#!/bin/bash
# Build a tuning variant of the library: scripts/build_variant.sh <name> <extra hipcc flags...>
# -> variants/libsrx_<name>.so (select it with SRX_LIBRARY=<path>)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/build/var_$name
mkdir -p "$obj" "$root/variants"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$root/include"
cd "$root/superresolution_amd/csrc"
/opt/rocm/bin/hipcc "$@" $F -c srx_kernels.hip -o "$obj/srx_kernels.o"
for f in srx_tvl1 srx_farneback srx_quality srx_solver srx_api; do
  [ -f "$root/build/srx/$f.o" ] || { echo "build the main library first"; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/variants/libsrx_$name.so" "$obj/srx_kernels.o" \
  "$root/build/srx/srx_tvl1.o" "$root/build/srx/srx_farneback.o" "$root/build/srx/srx_quality.o" "$root/build/srx/srx_solver.o" "$root/build/srx/srx_api.o"
echo "variants/libsrx_$name.so"
