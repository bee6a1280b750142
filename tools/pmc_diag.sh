#!/bin/bash
# Counter passes over the digit pass of tools/sort_once.py (2^28 pairs), one rocprofv3 run each:
#   tools/pmc_diag.sh OUTDIR [sort_once.py arguments...]
# MRS_LIBRARY_PATH selects a variant build.  Summarise with tools/pmc_summary.py OUTDIR.
set -e
out=$1; shift
mkdir -p "$out"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d "$out/$name" -o p --output-format csv -- \
      python3 tools/sort_once.py --log2n 28 --iters 2 $EXTRA_ARGS > "$out/$name.log" 2>&1
  echo "pass $name done"
}
EXTRA_ARGS="$*"
run tlb1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum
run tlb2 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run wr1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run wr2 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_BUBBLE_sum
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run grbm GRBM_GUI_ACTIVE
