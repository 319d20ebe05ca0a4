#!/bin/bash
# Instruction counts of the batched launch with phases switched off (debug flags): where the VALU instructions are.
# usage: bash tools/phase_pmc.sh <tag> [B]
TAG=${1:-r03}; B=${2:-64}
OUT=gpurun_out/phase_pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
for fl in 0 1 2 4 7; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
             "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $OUT/f${fl}_g$i -- python3 tools/batched_bench.py --B $B --steps 10 --warmup 2 --debug-flags $fl > $OUT/f${fl}_g$i.log 2>&1 || echo "group $i flags $fl failed"
  done
  python3 tools/pmc_summary.py --command "tools/batched_bench.py --B $B --debug-flags $fl" x $OUT/f${fl}_g* > $OUT/flags$fl.json
done
python3 - <<PY
import json
rows = {}
for fl in (0, 1, 2, 4, 7):
    d = json.load(open("$OUT/flags%d.json" % fl))
    rows[fl] = {k: v["mean"] for k, v in d.items() if k != "_meta"}
keys = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]
print("%-26s" % "counter (M per launch)" + "".join("%12s" % ("flags %d" % f) for f in rows))
for k in keys:
    print("%-26s" % k + "".join("%12.2f" % (rows[f].get(k, float("nan")) / 1e6) for f in rows))
json.dump(rows, open("gpurun_out/${TAG}_phase_pmc_B$B.json", "w"), indent=1)
PY
