# quick A/B (GPU box): 500-step bench lines of the headline workload (twice) and of one other workload; prints value, ms/step, kernel times
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 500 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2'); print(round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x['kernel'], round(x['avg_launch_us'],1))"
done
for wl in ${1:-community_small_CC}; do
python bench.py --no-cpu-baseline --workload $wl --steps 60 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['value'],1), round(d['ms_per_step'],4))"
done
