#!/bin/bash
# Diagnostic: per-phase instruction mix of k_xa / k_r2 (qm9_CC, B = 1024).
#   here:            bash tools/dev/phase_mix.sh build      (scratch library with -DCCSD_STOP_DIAG in tools/dev/_prof/, git-ignored)
#   on the GPU box:  bash tools/dev/phase_mix.sh run        (two rocprofv3 --pmc passes -> gpurun_out/phase_mix.txt)
# The diagnostic stamp() ends the grid at a chosen stamp; tools/dev/phase_mix.py launches the predictor once per stop point and the
# per-dispatch counters of successive stops are differenced into per-phase counts.
set -e
cd "$(dirname "$0")/../.."
R=$(pwd)
if [ "$1" = build ]; then
    O=ccsd_amd/csrc/_obj
    mkdir -p tools/dev/_prof
    for u in ccsd_xa ccsd_r2; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCCSD_STOP_DIAG -c ccsd_amd/csrc/$u.hip -o tools/dev/_prof/${u}_stop.o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/ccsd_hip.o tools/dev/_prof/ccsd_r2_stop.o $O/ccsd_r2b.o $O/ccsd_r2c.o $O/ccsd_r2d.o tools/dev/_prof/ccsd_xa_stop.o -o tools/dev/_prof/libccsd_stop.so
    rm tools/dev/_prof/*_stop.o
else
    export CCSD_LIB_PATH=$R/tools/dev/_prof/libccsd_stop.so      # (the product library stays in place: ccsd_amd/_lib.py honours the override)
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT -d $R/gpurun_out/pm1 -o run -- python3 $R/tools/dev/phase_mix.py launch > $R/gpurun_out/pm1.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $R/gpurun_out/pm2 -o run -- python3 $R/tools/dev/phase_mix.py launch > $R/gpurun_out/pm2.log 2>&1
    cd $R
    python3 tools/dev/phase_mix.py report gpurun_out/pm1 gpurun_out/pm2 > gpurun_out/phase_mix.txt
    cat gpurun_out/phase_mix.txt
fi
