for cfg in "0,0" "256,100" "256,200" "256,300" "1,200" "512,200" "768,100"; do
  CCSD_R2_STAGGER=$cfg python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2'); print('$cfg', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x['kernel'], round(x['avg_launch_us'],1))"
done
