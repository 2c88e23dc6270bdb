for cfg in "0,0" "1,100" "1,200" "2,100" "2,200" "256,100" "256,200" "512,100" "512,200" "3,100" "768,100"; do
  CCSD_XA_STAGGER=$cfg python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2'); print('$cfg', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x['kernel'], round(x['avg_launch_us'],1))"
done
