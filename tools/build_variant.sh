#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_FLAGS"  ->  gpurun_in/lib_NAME.so  (A/B builds of the kernel file; run from the repo root)
set -e
NAME=$1; EXTRA=$2
C=sr_livo_amd/csrc
mkdir -p gpurun_in /tmp/var_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-result -Wno-unused-value $EXTRA -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -c $C/srl_kernels.hip -o /tmp/var_$NAME/srl_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libsrlivo_hip.so -o gpurun_in/lib_$NAME.so /tmp/var_$NAME/srl_kernels.o $C/build/srl_map_kernels.o $C/build/srl_frame_kernels.o $C/build/srl_capi.o $C/build/srl_rccl.o \
    $C/build/eskfEstimator.o $C/build/lioOptimization.o $C/build/srl_host_capi.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built gpurun_in/lib_$NAME.so
