// Microbenchmark (diagnostic, run on the GPU box): mlp_chain_tile<2,4,1> (the final MLP of ScoreNetworkA_CC for qm9_CC:
// 30 -> 60 -> 60 -> 1 on 16-pair tiles) outside k_xa: 256-thread workgroups with 40 KB of LDS, 3 of 4 waves active.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iccsd_amd/csrc -DCCSD_BARRIER_PROF -o /tmp/ctile tools/ubench/chain_tile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ccsd_dev.h"

template <int NI, int NH, int NA>
__global__ __launch_bounds__(256, 4) void k(const MlpD* mp, const float* __restrict__ wp, float* out, long long* cyc, int reps) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 10240; i += 256) sm[i] = 0.001f * (i & 255);
    __builtin_amdgcn_s_barrier();
    const MlpD& m = *mp;
    const int wave = threadIdx.x >> 6;
    float acc = 0.f;
    if (wave < NA) {
        for (int r = 0; r < reps; ++r) {
            mlp_chain_tile<NI, NH, 1>(m, wp, sm, 96, sm, m.in, 16 * wave, 36, [](int row) { return row; },
                                    [&](int row, int f, float v) { sm[9000 + row] = v; acc += v; });
            if (blockIdx.x == 0 && threadIdx.x == 0)
                for (int q = 0; q < 5; ++q) cyc[r * 8 + q] = g_ct[q + 1] - g_ct[q];
        }
    }
    __builtin_amdgcn_s_barrier();
    out[blockIdx.x * 256 + threadIdx.x] = acc + sm[9000 + (threadIdx.x & 31)];
}

template <int NI, int NH, int NA>
void run(const char* name, int in, int hid, int outw) {
    MlpD m{};
    m.n = 3; m.in = in; m.hid = hid; m.out = outw; m.chain = 3;
    int pcur = 0;
    for (int i = 0; i < 3; ++i) { const int ip = 16 * (i == 0 ? NI : NH), op = 16 * (i == 2 ? 1 : NH); m.pw[i] = pcur; pcur += op * ip; m.pb[i] = pcur; pcur += op; }
    std::vector<float> hw(pcur, 0.01f);
    MlpD* dm; float *w, *out; long long* cyc;
    hipMalloc(&dm, sizeof(MlpD)); hipMalloc(&w, pcur * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 64 * 8 * 8);
    hipMemcpy(dm, &m, sizeof(MlpD), hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), pcur * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) { k<NI, NH, NA><<<blocks, 256, 40960>>>(dm, w, out, cyc, 3); hipDeviceSynchronize(); }
        long long h[64 * 8];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        for (int r = 0; r < 3; ++r)
            printf("%s, %d active waves, blocks %4d  call %d: gather %lld, linear 1 %lld, middle %lld, last %lld, epilogue %lld\n", name, NA, blocks, r, h[r * 8], h[r * 8 + 1], h[r * 8 + 2], h[r * 8 + 3], h[r * 8 + 4]);
    }
}
int main() {
    run<2, 4, 3>("final MLP <2,4,1> 30-60-60-1", 30, 60, 1);
    run<2, 3, 1>("X head <2,3,1> 24-48-48-4", 24, 48, 4);
    run<1, 1, 3>("edge MLP <1,1,1> 16-16-16-8", 16, 16, 8);
    return 0;
}
