#!/bin/bash
# Kernel trace + PMC passes of ONE command (the throughput regimes: batched B = 64, config 3), summaries into gpurun_out/.
# usage: bash tools/profile_regime.sh <tag> <commit> <name> <min_grid> <python args...>
#   e.g. bash tools/profile_regime.sh r03 abc1234 B64 0 tools/batched_bench.py --B 64
# Counters are collected in their own runs (--pmc only), one small group per run, so an unknown name costs its group only.
TAG=$1; COMMIT=$2; NAME=$3; MING=$4; shift 4
OUT=gpurun_out/prof_${TAG}_$NAME
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/trace.log; exit 1; }
ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$ST" gpurun_out/${TAG}_kernel_stats_$NAME.csv
sed -i "1s|^|# commit $COMMIT; rocprofv3 --kernel-trace --stats -- python3 $*\n|" gpurun_out/${TAG}_kernel_stats_$NAME.csv
tail -2 $OUT/trace.log
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
  "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" \
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
  "SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 "$@" > $OUT/pmc$i.log 2>&1 || echo "pmc group '$grp' failed (see $OUT/pmc$i.log): $(grep -i -m1 'error\|invalid\|not' $OUT/pmc$i.log)"
done
python3 tools/pmc_summary.py --command "python3 $*" --min-grid $MING $COMMIT $OUT/pmc* > gpurun_out/${TAG}_pmc_summary_$NAME.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_pmc_summary_$NAME.json"))
print({k: (round(v["mean"], 1) if isinstance(v, dict) and "mean" in v else v) for k, v in d.items() if k != "_meta"})
print(d["_meta"].get("dispatch"))
PY
head -5 gpurun_out/${TAG}_kernel_stats_$NAME.csv
