// Microbenchmark (diagnostic, run on the GPU box): issue cost of the integer multiplies Philox4x32 needs on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/int_mul_rate tools/ubench/int_mul_rate.hip && /tmp/int_mul_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(unsigned int* out, int iters) {
    unsigned int a = threadIdx.x * 2654435761u + 1u, b = a ^ 0x9E3779B9u, c = b + 7u, d = c ^ 0x85EBCA6Bu;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {          // v_mad_u64_u32 (what the compiler emits for (u64)K * x)
                unsigned long long p, q;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(0xD2511F53u) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"(c), "v"(0xCD9E8D57u) : "vcc");
                a = (unsigned int)(q >> 32) ^ b; b = (unsigned int)q; c = (unsigned int)(p >> 32) ^ d; d = (unsigned int)p;
            } else if (MODE == 1) {   // v_mul_hi_u32 + v_mul_lo_u32
                unsigned int h0, l0, h1, l1;
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(h0) : "v"(a), "v"(0xD2511F53u));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(l0) : "v"(a), "v"(0xD2511F53u));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(h1) : "v"(c), "v"(0xCD9E8D57u));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(l1) : "v"(c), "v"(0xCD9E8D57u));
                a = h1 ^ b; b = l1; c = h0 ^ d; d = l0;
            } else if (MODE == 2) {   // reference: 4 dependent-free full-rate ops per round (xor / add)
                a = (a ^ b) + 0x9E3779B9u; b = (b ^ c) + 0x85EBCA6Bu; c = (c ^ d) + 0xC2B2AE35u; d = (d ^ a) + 0x27D4EB2Fu;
            } else {                  // 24-bit multiplies (full rate?): v_mul_u32_u24 + v_mul_hi_u32_u24
                unsigned int h0, l0;
                asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(h0) : "v"(a), "v"(0x511F53u));
                asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(l0) : "v"(a), "v"(0x511F53u));
                unsigned int h1, l1;
                asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(h1) : "v"(c), "v"(0x9E8D57u));
                asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(l1) : "v"(c), "v"(0x9E8D57u));
                a = h1 ^ b; b = l1; c = h0 ^ d; d = l0;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int MODE>
static void run(const char* name, unsigned int* out) {
    const int iters = 4096, blocks = 256 * 8, threads = 256;     // 8 waves per SIMD: issue-bound
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks*threads/64 waves over 1024 SIMDs, iters*8 rounds each
    const double rounds_per_simd = (double)blocks * threads / 64 / 1024 * iters * 8;
    printf("%-34s %8.3f ms  -> %6.1f ns per round per SIMD (x 2.4 GHz = %6.1f cycles)\n", name, ms, ms * 1e6 / rounds_per_simd,
           ms * 1e6 / rounds_per_simd * 2.4);
}

int main() {
    unsigned int* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<2>("4 x (xor + add), full rate", out);
    run<0>("2 x v_mad_u64_u32 + 2 xor", out);
    run<1>("2 x (v_mul_hi_u32 + v_mul_lo_u32)", out);
    run<3>("2 x (mul_hi_u24 + mul_u24)", out);
    hipFree(out);
    return 0;
}
