#!/bin/bash
# dev: recompile ONE translation unit with extra flags and link a side copy of the library (for VDN_LIB=... A/B runs)
#   tools/dev/relink.sh sdf_lw_bf16 out_name [extra hipcc flags]
set -e
cd /root/repo/vdn-nerf_amd
TU=$1; OUT=$2; shift 2
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../include -I csrc"
case $TU in sdf_bf16|sdf_lw_bf16) FL="$FL -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1";; esac
mkdir -p ../gpurun_tmp
hipcc $FL "$@" -c csrc/$TU.hip -o ../gpurun_tmp/$OUT.o
OBJS=$(ls vdn_hip/_build/*.o | grep -v "/$TU.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o vdn_hip/libvdn_render_$OUT.so $OBJS ../gpurun_tmp/$OUT.o
echo built vdn_hip/libvdn_render_$OUT.so
