for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 300 --warmup 30 --no-kernel-timing 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
m = d['closed_loop']['modes']
print('GPU_MAX_HW_QUEUES=$q', round(d['ms_per_step']*1e3,2), round(d['two_steps_in_flight']['ms_per_step']*1e3,2), {k: {mm: round(v[mm]['us_per_step'], 2) for mm in v} for k, v in m.items()})"
done
