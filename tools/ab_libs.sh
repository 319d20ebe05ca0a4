#!/bin/bash
# usage: bash tools/ab_libs.sh <configs> <lib name> [<lib name> ...]   -- perf_suite per library (<pkg>/lib/<name>), two interleaved rounds
P=$(ls -d catenary-*_amd)
ONLY=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    ROVMPC_LIB_OLD_ABI=1 ROVMPC_LIB=$P/lib/$L python3 tools/perf_suite.py --tag ${L#librovmpc_} --only $ONLY 2>/dev/null | grep -v "^{"
  done
done
