timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log; python bench.py --steps 200 --warmup 20 2>/dev/null > gpurun_out/b.json && python bench.py --workload community_small_CC --steps 100 --warmup 10 2>/dev/null > gpurun_out/b_cs.json; python - <<EOF
import json
for f in ("gpurun_out/b.json","gpurun_out/b_cs.json"):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["avg_launch_us"], j.get("roofline_k_xa",{}).get("avg_launch_us"))
    except Exception as e: print(f, "ERR", e)
EOF
