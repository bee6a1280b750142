#!/bin/bash
# Builds scratch/libmrs_diag.so: the library with -DMRS_STAMPS (phase clocks per tile).
exec "$(dirname "$0")/build_variant.sh" diag -DMRS_STAMPS $EXTRA
