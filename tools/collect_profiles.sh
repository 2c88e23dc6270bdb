#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <tag> pmc|bench [workload]
#   pmc    rocprofv3 --kernel-trace --stats of a 200-step bench run + the three --pmc passes (each its own run) -> gpurun_out/
#          (workload: default qm9_CC; e.g. community_small_CC -> 30-step stats run, passes tagged <workload>_fetch ...)
#   bench  the bench lines (default workload with the CPU baseline, then the other BASELINE workloads) -> gpurun_out/<tag>_*.json
# Afterwards, here: python tools/pmc_traffic.py <round> fetch write sq; copy the summaries into profiles/ (profiles/README.md).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
cd /tmp && export TMPDIR=/tmp
if [ "$2" = pmc ]; then
    wl=${3:-qm9_CC}
    pre=""; steps=1000
    if [ "$wl" != qm9_CC ]; then pre="${wl}_"; steps=30; fi
    export PMC_WORKLOAD=$wl
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$wl -o run -- python3 $R/bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-kernel-events --warmup-seconds 0 > $R/gpurun_out/prof_${tag}_$wl.log 2>&1
    echo "stats done"
    cd $R
    bash tools/pmc_run.sh ${pre}fetch FETCH_SIZE > /dev/null && echo "fetch done"
    bash tools/pmc_run.sh ${pre}write WRITE_SIZE > /dev/null && echo "write done"
    bash tools/pmc_run.sh ${pre}sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT > /dev/null && echo "sq done"
else
    cd $R
    python3 bench.py 2> gpurun_out/${tag}_bench.err > gpurun_out/${tag}_bench.json && echo "qm9_CC done"
    python3 bench.py --steps 20 --warmup 5 2> /dev/null > gpurun_out/${tag}_driver_bench.json && echo "qm9_CC, the driver's 20-step line done"
    python3 bench.py --batch 2500 --steps 200 --no-cpu-baseline 2> /dev/null > gpurun_out/${tag}_batch2500_bench.json && echo "qm9_CC at the shipped YAML's chunk (2500) done"
    # BASELINE configs[1] with its roofline and CPU-baseline legs (the other workloads: kernel-trace summaries only)
    python3 bench.py --workload community_small_CC --steps 60 --warmup 5 2> /dev/null > gpurun_out/${tag}_community_small_CC_full_bench.json && echo "community_small_CC (roofline + cpu baseline) done"
    # the split-precision experiment (never the default): its own labelled line
    python3 bench.py --workload community_small_CC --steps 60 --warmup 5 --no-cpu-baseline --split-bf16 3 2> /dev/null > gpurun_out/${tag}_community_small_CC_bf16x3_bench.json && echo "community_small_CC, bf16 x 3 experiment done"
    for wl in community_small_CC zinc250k_CC_5b enzymes_small_CC qm9_Base_CC zinc250k community_small; do
        cd /tmp
        rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$wl -o run -- python3 $R/bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-events --warmup-seconds 0 > $R/gpurun_out/${tag}_${wl}_bench.json 2> $R/gpurun_out/${tag}_${wl}.err || echo "$wl failed"
        echo "$wl done"
    done
fi
