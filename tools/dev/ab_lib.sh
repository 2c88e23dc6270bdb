# A/B (GPU box): bench the headline workload with the product library, then with tools/dev/_prof/libccsd_exp.so selected through CCSD_LIB_PATH.  Extra environment for both runs: AB_ENV="CCSD_NO_BAKE=1" bash tools/dev/ab_lib.sh
run() { env $AB_ENV python bench.py --no-cpu-baseline --steps 500 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; x=d.get('roofline_k_xa') or d.get('roofline_k_r2'); print('$1', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['avg_launch_us'],1), x['kernel'], round(x['avg_launch_us'],1))"; }
run product; run product
export CCSD_LIB_PATH=$PWD/tools/dev/_prof/libccsd_exp.so      # (the product library stays in place)
run experiment; run experiment
