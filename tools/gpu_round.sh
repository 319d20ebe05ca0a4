#!/bin/bash
# One gpurun call: GPU tests (optionally -k filtered), default bench summary, optional extra commands.
# usage: bash tools/gpu_round.sh ["pytest -k expr" | all | none] [extra command ...]
mkdir -p gpurun_out
export TMPDIR=/tmp
K=${1:-all}; shift
if [ "$K" != "none" ]; then
  if [ "$K" = "all" ]; then KARG=(); else KARG=(-k "$K"); fi
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q "${KARG[@]}" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("headline us", d["ms_per_step"] * 1e3, "batched", [(r["B"], round(r["value"] / 1e9, 2)) for r in d["batched"]["runs"]])
print("closed_loop", json.dumps(d.get("closed_loop", {}).get("modes")), "2-in-flight us", d.get("two_steps_in_flight", {}).get("ms_per_step"))
PY
for cmd in "$@"; do echo "== $cmd"; timeout -k 10 600 bash -c "$cmd" 2>&1 | tail -12; done
