#!/bin/bash
# quick variant: only the u32 radix units + api differ; reuse other objects from the main build
name=$1; shift
cd /root/repo/wgpu_morton_code_radix_sort_test_amd/csrc
obj=/tmp/mrs_variant_$name; rm -rf $obj; mkdir -p $obj ../../scratch
for f in mrs_radix_u32 mrs_radix_u32_atomic; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../include -I. "$@" -c $f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libmrs_$name.so $obj/*.o mrs_api.o mrs_hybrid.o mrs_split.o mrs_dist.o mrs_radix_u64.o mrs_radix_u64_atomic.o mrs_scan.o mrs_morton.o mrs_probe.o mrs_small_sort.o mrs_gather.o -ldl
