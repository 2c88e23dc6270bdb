# A/B by environment (GPU box): bash tools/dev/ab_env.sh "<env settings A>" "<env settings B>" ...   (use "-" for no extra environment)
# One 500-step bench line of the headline workload per setting, twice round-robin (clock drift shows as the spread between the rounds).
run() { env $1 python bench.py --no-cpu-baseline --steps ${AB_STEPS:-500} ${AB_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2') or {}; print('%-44s' % '$1', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x.get('kernel'), round(x.get('avg_launch_us',0),1))"; }
for rep in 1 2; do
for e in "$@"; do
    if [ "$e" = "-" ]; then run ""; else run "$e"; fi
done
done
