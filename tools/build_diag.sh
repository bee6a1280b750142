#!/bin/bash
# Builds scratch/libmrs_diag.so: the library with -DMRS_STAMPS (phase clocks per tile).
set -e
cd "$(dirname "$0")/../wgpu_morton_code_radix_sort_test_amd/csrc"
mkdir -p /tmp/diag ../../scratch
for f in mrs_api mrs_radix_u32 mrs_radix_u32_atomic mrs_radix_u64 mrs_radix_u64_atomic mrs_scan mrs_morton mrs_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
    -fhip-fp32-correctly-rounded-divide-sqrt -I../../include -I. -DMRS_STAMPS $EXTRA -c $f.hip -o /tmp/diag/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libmrs_diag.so /tmp/diag/*.o
