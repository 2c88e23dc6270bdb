#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag> pmc|bench
#   pmc    rocprofv3 --kernel-trace --stats of a 200-step bench run + the three --pmc passes (each its own run) -> gpurun_out/
#   bench  the bench lines (default workload with the CPU baseline, then the other BASELINE workloads) -> gpurun_out/<tag>_*.json
# Afterwards, here: python tools/pmc_traffic.py <round> fetch write sq; copy the summaries into profiles/ (profiles/README.md).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
cd /tmp && export TMPDIR=/tmp
if [ "$2" = pmc ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/prof_$tag.log 2>&1
    echo "stats done"
    cd $R
    bash tools/pmc_run.sh fetch FETCH_SIZE > /dev/null && echo "fetch done"
    bash tools/pmc_run.sh write WRITE_SIZE > /dev/null && echo "write done"
    bash tools/pmc_run.sh sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT > /dev/null && echo "sq done"
else
    cd $R
    python3 bench.py 2> gpurun_out/${tag}_bench.err > gpurun_out/${tag}_bench.json && echo "qm9_CC done"
    for wl in community_small_CC zinc250k_CC_5b enzymes_small_CC qm9_Base_CC zinc250k community_small; do
        cd /tmp
        rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$wl -o run -- python3 $R/bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/${tag}_${wl}_bench.json 2> $R/gpurun_out/${tag}_${wl}.err || echo "$wl failed"
        echo "$wl done"
    done
fi
