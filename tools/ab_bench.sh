# usage: bash tools/ab_bench.sh libA.so libB.so  -- alternating bench runs of two builds on the same box (box-to-box spread is +-0.3 us);
# put the other build at <pkg>/lib/libA.so (e.g. git archive HEAD | build -> librovmpc_prev.so)
P=$(ls -d catenary-*_amd)
for rep in 1 2; do for L in "$@"; do ROVMPC_LIB_OLD_ABI=1 ROVMPC_LIB=$P/lib/$L timeout -k 10 120 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-kernel-timing --no-pipelined-extra --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$L\", round(d[\"ms_per_step\"]*1000,3))"; done; done
