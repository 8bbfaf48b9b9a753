#!/bin/bash
# builds tools/dev/sdf2_bench (development harness of csrc/k_sdf_fwd2.h): one object per line of variants.txt, in parallel
set -e
cd "$(dirname "$0")"
ROOT=../..
OUT=_build; mkdir -p $OUT; rm -f $OUT/v_*.o
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -I $ROOT/include -I $ROOT/vdn-nerf_amd/csrc"
: > sdf2_variants.inc
pids=()
vid=0
while read -r tag m s n d extra; do
  case "$tag" in ''|\#*) continue;; esac
  echo "V2($tag, $m, $s)" >> sdf2_variants.inc
  vid=$((vid+1))
  hipcc $FLAGS -DVARIANT_ID=$vid -DVTAG=$tag -DVM=$m -DVS=$s -DVN=$n -DVD=$d $extra -c sdf2_variant.hip -o $OUT/v_$tag.o & pids+=($!)
  while [ $(jobs -r | wc -l) -ge ${JOBS:-8} ]; do sleep 1; done
done < variants.txt
for p in "${pids[@]}"; do wait $p; done
hipcc $FLAGS -c sdf2_bench.hip -o $OUT/main.o
hipcc --offload-arch=gfx950 -o sdf2_bench $OUT/*.o -L $ROOT/vdn-nerf_amd/vdn_hip -lvdn_render -Wl,-rpath,'$ORIGIN/../../vdn-nerf_amd/vdn_hip'
echo built tools/dev/sdf2_bench
