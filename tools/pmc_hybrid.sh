#!/bin/bash
# SQ and GRBM counter passes over the kernels of the MRS_ALGO_HYBRID schedule, one rocprofv3 run each
# (2^28 pairs; extra arguments go to tools/sort_once.py, e.g. --dist surface):
#   tools/pmc_hybrid.sh OUTDIR [sort_once.py arguments...]
out=$1; shift
mkdir -p "$out"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d "$out/$name" -o p --output-format csv -- \
      python3 tools/sort_once.py --log2n 28 --iters 2 --algo hybrid --end-bit 30 $EXTRA_ARGS > "$out/$name.log" 2>&1 \
    || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
  echo "pass $name done"
}
EXTRA_ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run grbm GRBM_GUI_ACTIVE &&
echo "hybrid counters collected"
