#!/bin/bash
# Diagnostic: link a scratch library whose ccsd_xa unit is built with -DCCSD_BARRIER_PROF (tools/dev/_prof/, git-ignored).
#   here:            bash tools/dev/barrier_prof.sh build
#   on the GPU box:  bash tools/dev/barrier_prof.sh run [B]     (selected through CCSD_LIB_PATH: the product library stays in place)
# Prints, per wave of k_xa, the share of its life spent waiting in __syncthreads().
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
    O=ccsd_amd/csrc/_obj
    mkdir -p tools/dev/_prof
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCCSD_BARRIER_PROF -c ccsd_amd/csrc/ccsd_xa.hip -o tools/dev/_prof/ccsd_xa_prof.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/ccsd_hip.o $O/ccsd_r2.o $O/ccsd_r2b.o $O/ccsd_r2c.o $O/ccsd_r2d.o tools/dev/_prof/ccsd_xa_prof.o -o tools/dev/_prof/libccsd_hip.so
    rm tools/dev/_prof/ccsd_xa_prof.o
else
    CCSD_LIB_PATH=$PWD/tools/dev/_prof/libccsd_hip.so STAMPS_BARRIERS=1 python tools/stamps.py "${2:-1024}"
fi
