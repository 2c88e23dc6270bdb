// Microbenchmark (diagnostic, run on the GPU box): what does one linear of the register-resident MLP chain cost a lone wave?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iccsd_amd/csrc -o /tmp/chain tools/ubench/chain_layer.hip && /tmp/chain
// One 64-thread workgroup per CU slot runs chain_layer<4,4> (64 x 64 linear, 16 rows: 64 MFMAs, 16 float4 weight loads per
// lane, ELU) REP times back to back on warm caches and reports cycles per call, for:
//   A  chain_layer as in ccsd_dev.h (weights loaded tile by tile, the compiler's interleaving)
//   B  all 16 float4 of the lane loaded up front, then the MFMAs
//   C  MFMAs only (weights in registers, loaded once outside the loop)
//   D  as A without the ELU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ccsd_dev.h"

constexpr int REP = 64;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ wp, float* out, long long* cyc, int nactive) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    if ((int)(threadIdx.x >> 6) >= nactive) { __syncthreads(); return; }
    chain_f32x4 h0[4], h1[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) h0[t][j] = 0.01f * (lane + t + j);
    const float* W = wp + 4 * lane;
    const float* Bv = wp + 4096 + 4 * kq;
    float4 wv[4][4];
    if (MODE == 2) for (int to = 0; to < 4; ++to) for (int t = 0; t < 4; ++t) wv[to][t] = *reinterpret_cast<const float4*>(W + (size_t)(16 * to) * 64 + 16 * t);
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) chain_layer<4, 4>(W, Bv, true, h0, h1);
        if (MODE == 3) chain_layer<4, 4>(W, Bv, false, h0, h1);
        if (MODE == 1 || MODE == 2) {
            if (MODE == 1) {
#pragma unroll
                for (int to = 0; to < 4; ++to)
#pragma unroll
                    for (int t = 0; t < 4; ++t) wv[to][t] = *reinterpret_cast<const float4*>(W + (size_t)(16 * to) * 64 + 16 * t);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                chain_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[to][t].x, h0[t][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[to][t].y, h0[t][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[to][t].z, h0[t][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[to][t].w, h0[t][3], acc, 0, 0, 0);
                }
                for (int j = 0; j < 4; ++j) acc[j] = elu1_sel(acc[j]);
                h1[to] = acc;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) h0[t] = h1[t] * 0.5f;
        asm volatile("" ::: "memory");
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) s += h0[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / REP;
    __syncthreads();
}

int main() {
    std::vector<float> hw(4096 + 64, 0.01f);
    float *w, *out; long long* cyc;
    hipMalloc(&w, hw.size() * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"A chain_layer (ELU)", "B loads up front", "C weights in registers", "D chain_layer (no ELU)"};
    for (int blocks : {256, 1024}) {
        for (int threads : {64, 256}) {
            for (int nactive = 1; nactive <= threads / 64; ++nactive) {
                for (int mode = 0; mode < 4; mode += 2) {
                    for (int rep = 0; rep < 2; ++rep) {
                        if (mode == 0) k<0><<<blocks, threads>>>(w, out, cyc, nactive);
                        if (mode == 2) k<2><<<blocks, threads>>>(w, out, cyc, nactive);
                        hipDeviceSynchronize();
                    }
                    std::vector<long long> h(blocks);
                    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
                    long long s = 0; for (auto v : h) s += v;
                    printf("blocks %4d x %3d threads, %d waves active  %-26s %6lld cycles per 64x64 linear (64 MFMAs), wave 0\n", blocks, threads, nactive, names[mode], s / blocks);
                }
            }
        }
    }
    return 0;
}
