#!/bin/bash
# rocprofv3 evidence for the MRS_ALGO_HYBRID schedule: tools/collect_hybrid_profiles.sh TAG
# (kernel-trace statistics of the default bench.py run and of one sort per size; FETCH_SIZE and
# WRITE_SIZE of its kernels in runs of their own).  Raw output: gpurun_out/TAG/; condensed by
# tools/summarize_profiles.py --hybrid-root gpurun_out/TAG into profiles/.
tag=${1:-r02_hybrid}
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
stats hybrid_bench python3 bench.py --steps 10 --warmup 3 --no-extras &&
stats hybrid_2p29 python3 tools/sort_once.py --log2n 29 --iters 4 --algo hybrid --end-bit 30 &&
stats hybrid_2p30 python3 tools/sort_once.py --log2n 30 --iters 4 --algo hybrid --end-bit 30 &&
stats hybrid_cluster python3 tools/sort_once.py --log2n 28 --iters 4 --algo hybrid --end-bit 30 --dist cluster5 &&
stats hybrid_2p26 python3 tools/sort_once.py --log2n 26 --iters 4 --algo hybrid --end-bit 30 &&
stats hybrid_surface python3 tools/sort_once.py --log2n 28 --iters 4 --algo hybrid --end-bit 30 --dist surface &&
pmc fetch FETCH_SIZE python3 tools/sort_once.py --log2n 28 --iters 2 --algo hybrid --end-bit 30 &&
pmc write WRITE_SIZE python3 tools/sort_once.py --log2n 28 --iters 2 --algo hybrid --end-bit 30 &&
echo "all hybrid profiles collected"
