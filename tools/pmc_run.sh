#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <tag> <counter> [<counter> ...]   -> gpurun_out/pmc_<tag>/
# (PMC_WORKLOAD=community_small_CC selects another bench workload.)  One rocprofv3 --pmc pass (kernel-trace only, its own run) over a short bench.py run; per-kernel means printed as JSON.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/pmc_$tag -o run -- python3 $R/bench.py --workload ${PMC_WORKLOAD:-qm9_CC} --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --warmup-seconds 0 > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_$tag
