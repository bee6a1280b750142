#!/bin/bash
# Builds a variant of the library for A/B runs: tools/build_variant.sh NAME [-DFLAG ...]
# writes scratch/libmrs_NAME.so (use it through MRS_LIBRARY_PATH).
set -e
name=$1; shift
cd "$(dirname "$0")/../wgpu_morton_code_radix_sort_test_amd/csrc"
obj=/tmp/mrs_variant_$name
rm -rf $obj; mkdir -p $obj ../../scratch
for f in mrs_api mrs_hybrid mrs_split mrs_dist mrs_radix_u32 mrs_radix_u32_atomic mrs_radix_u64 mrs_radix_u64_atomic mrs_scan mrs_morton mrs_probe mrs_small_sort mrs_gather; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
    -fhip-fp32-correctly-rounded-divide-sqrt -I../../include -I. "$@" -c $f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libmrs_$name.so $obj/*.o -ldl
