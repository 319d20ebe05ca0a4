#!/bin/bash
# A/B of two builds on one box: perf_suite with <pkg>/lib/librovmpc_base.so and with the in-tree library, interleaved twice.
# usage: bash tools/ab_suite.sh [configs]      (configs: --only list of tools/perf_suite.py)
P=$(ls -d catenary-*_amd)
ONLY=${1:-c2,b64,c3,k32k,jit,r59}
for rep in 1 2; do
  ROVMPC_LIB_OLD_ABI=1 ROVMPC_LIB=$P/lib/librovmpc_base.so python3 tools/perf_suite.py --tag base --only $ONLY 2>/dev/null | grep -v "^{"
  python3 tools/perf_suite.py --tag new --only $ONLY 2>/dev/null | grep -v "^{"
done
