for m in 1 0 2; do
  CCSD_XA_PRIO=$m python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2'); print('prio mode $m', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x['kernel'], round(x['avg_launch_us'],1))"
done
python tools/stamps.py 2>&1 | grep "k_xa workgroup life\|co-resident" -A2 | head -6
