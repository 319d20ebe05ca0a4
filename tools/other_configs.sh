#!/bin/bash
# Bench lines of the other sizes / models (one JSON line each) -> gpurun_out/<tag>_other_configs.jsonl
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_other_configs.jsonl
: > $OUT
B="python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-kernel-timing --no-pipelined-extra --no-extras"
$B --K 32768 2>/dev/null >> $OUT
$B --K 262144 --steps 60 --warmup 10 2>/dev/null >> $OUT
$B --N 50 --K 16384 --dtype f32 2>/dev/null >> $OUT
for m in jit-default rows:5,9 gen2 gen3 rows:30,27; do $B --model $m 2>/dev/null >> $OUT; done
$B --interp --steps 60 --warmup 10 2>/dev/null >> $OUT
python - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l); c = d["config"]
    print(c["model"], c["N"], c["K_per_gpu"], d["dtype"], round(d["ms_per_step"] * 1e3, 2), "us", "%.3g" % d["value"])
PY
