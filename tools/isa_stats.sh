#!/bin/bash
# device assembly + per-kernel register / scratch / size summary:  tools/isa_stats.sh [pattern]
mkdir -p /tmp/isa && cd /tmp/isa && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o ccsd.s /root/repo/ccsd_amd/csrc/ccsd_hip.hip 2>/dev/null
awk '/^_Z.*:/{name=$1; start=NR} /^; NumVgprs:/{v=$3} /^; ScratchSize:/{sc=$3} /^; Occupancy:/{printf "%-70s lines %6d vgpr %3d scratch %4d occ %d\n", substr(name,1,70), NR-start, v, sc, $3}' ccsd.s | grep "${1:-.}"
