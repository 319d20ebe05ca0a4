#!/bin/bash
# Round profile of the bench command on the GPU box: kernel trace + stats, then PMC passes (one counter
# group per run: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), summaries into gpurun_out/.
# usage: bash tools/profile_round.sh <tag> <commit>      (copy gpurun_out/<tag>_* into profiles/ afterwards)
set -e
TAG=${1:-r02}
COMMIT=${2:-unknown}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-kernel-timing --no-pipelined-extra --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
sed -i "1s/^/# commit $COMMIT; rocprofv3 --kernel-trace --stats -- python3 $ARGS\n/" gpurun_out/${TAG}_kernel_stats.csv
PARGS="bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-pipelined-extra --no-extras"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 $PARGS > $OUT/pmc$i.log 2>&1 || echo "pmc group '$grp' failed (see $OUT/pmc$i.log)"
done
python3 tools/pmc_summary.py $COMMIT $OUT/pmc* > gpurun_out/${TAG}_pmc_summary.json
cp gpurun_out/${TAG}_pmc_summary.json profiles/${TAG}_pmc_summary.json     # so that the bench run below can quote it
python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> $OUT/bench.log
tail -c 3000 gpurun_out/${TAG}_bench_default.json
head -4 gpurun_out/${TAG}_kernel_stats.csv
