#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_HIPCC_FLAGS" -- A/B build of the device library into
# gpurun_out-independent path ngmlr_amd/variants/libcvxalign_NAME.so (tuning aid).
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$HERE/ngmlr_amd/variants" /tmp/cvx_variant_$1
cd "$HERE/ngmlr_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC -I../../include -I. $2 -c cvx_kernels.hip -o /tmp/cvx_variant_$1/k.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/ngmlr_amd/variants/libcvxalign_$1.so" /tmp/cvx_variant_$1/k.o $(ls build/*.o | grep -v cvx_kernels.o) -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
echo built $1
