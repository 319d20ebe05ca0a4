#!/bin/bash
# One gpurun call: selected GPU tests, then the default bench.  usage: bash tools/gpu_round.sh [pytest -k expression]
mkdir -p gpurun_out
export TMPDIR=/tmp
K=${1:-}
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
fi
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("headline us", d["ms_per_step"] * 1e3, "batched", [(r["B"], round(r["value"] / 1e9, 2)) for r in d["batched"]["runs"]])
print("closed_loop", json.dumps(d.get("closed_loop", {}).get("modes")))
PY
