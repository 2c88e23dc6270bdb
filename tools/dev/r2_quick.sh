# quick A/B of k_r2 (GPU box): 300-step bench line + phase stamps of a predictor and a norms launch
python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2'); print(round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x['kernel'], round(x['avg_launch_us'],1))"
python tools/dev/stamps_r2.py pred 2>&1 | grep -A7 "first 512"
python tools/dev/stamps_r2.py norms 2>&1 | grep -A7 "first 512"
