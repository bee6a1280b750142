#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box: tools/collect_profiles.sh TAG
# (kernel-trace statistics of the benchmark and of the other shapes; FETCH_SIZE / WRITE_SIZE of the
# digit pass in runs of their own).  Raw output: gpurun_out/TAG/; condensed by
# tools/summarize_profiles.py into profiles/.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
stats() {  # stats NAME command...
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $out/prof_$name -o p --output-format csv -- "$@" > $out/prof_$name.log 2>&1 \
    || { echo "profile $name failed"; tail -5 $out/prof_$name.log; return 1; }
  echo "profile $name done"
}
pmc() {  # pmc NAME COUNTER command...
  name=$1; counter=$2; shift 2
  timeout -s KILL 200 rocprofv3 --pmc $counter -d $out/pmc_$name -o p --output-format csv -- "$@" > $out/pmc_$name.log 2>&1 \
    || { echo "pmc $name failed"; tail -5 $out/pmc_$name.log; return 1; }
  echo "pmc $name done"
}
stats bench python3 bench.py --steps 5 --warmup 2 &&
stats three_kernel python3 tools/sort_once.py --log2n 28 --iters 3 --algo three_kernel &&
stats u64_tri16 python3 tools/sort_once.py --log2n 27 --iters 3 --key-words 2 --payload-bytes 16 &&
stats u64_idx python3 tools/sort_once.py --log2n 27 --iters 3 --key-words 2 --payload-bytes 4 &&
stats partition python3 tools/partition_bench.py &&
stats encode_lut python3 tools/encode_lut_bench.py &&
MRS_BENCH_FORCE_DIST=1 stats dist1 python3 bench.py --steps 3 --warmup 1 &&
pmc fetch FETCH_SIZE python3 tools/sort_once.py --log2n 28 --iters 2 &&
pmc write WRITE_SIZE python3 tools/sort_once.py --log2n 28 --iters 2 &&
echo "all profiles collected"
