#!/bin/bash
# One gpurun call: GPU test-suite, default bench, A/B against the previous build, gloo rehearsal of the N > 1 path.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench_default.json
bash tools/ab_bench.sh librovmpc_prev.so librovmpc.so 2>&1 | tee gpurun_out/ab.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --devices 0,0 --steps 200 --warmup 20 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err; echo "gloo2 rc=$?"
tail -c 1500 gpurun_out/bench_gloo2.json; tail -5 gpurun_out/bench_gloo2.err
