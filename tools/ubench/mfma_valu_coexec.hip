// Microbenchmark (diagnostic, run on the GPU box): do f32 MFMAs (v_mfma_f32_16x16x4_f32) and f32 VALU work overlap on one SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coexec tools/ubench/mfma_valu_coexec.hip && /tmp/coexec
// 256-thread workgroups, one per CU slot (4 waves = one per SIMD) or 512 threads (two per SIMD).  Roles by wave:
//   M = MFMA only, V = VALU fma only, I = integer VALU only (v_mad_u64_u32 + xor: Philox-like), X = MFMA and VALU interleaved in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_loop(int iters, float* sink) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
        }
    }
    *sink = a0[0] + a1[1] + a2[2] + a3[3];
}
__device__ __forceinline__ void valu_loop(int iters, float* sink) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float m = 1.000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], m, c);          // 128 independent-ish fmas per iteration
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    *sink = s;
}
__device__ __forceinline__ void int_loop(int iters, float* sink) {
    unsigned a = threadIdx.x * 2654435761u + 1u, b = a ^ 0x9E3779B9u, c = b + 7u, d = c ^ 0x85EBCA6Bu;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {                                    // 16 Philox-like rounds: 2 mad_u64 + 4 xor = 96 instr
            const unsigned long long p = (unsigned long long)0xD2511F53u * a, q = (unsigned long long)0xCD9E8D57u * c;
            a = (unsigned)(q >> 32) ^ b ^ 0x9E3779B9u; b = (unsigned)q; c = (unsigned)(p >> 32) ^ d ^ 0xBB67AE85u; d = (unsigned)p;
        }
    }
    *sink = (float)(a ^ b ^ c ^ d);
}
__device__ __forceinline__ void mixed_loop(int iters, float* sink) {     // per MFMA: 6 fmas, in one instruction stream
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    float v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f, m = 1.000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = fmaf(v[j], m, c);
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = fmaf(v[j], m, c);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = a0[0] + a1[1];
#pragma unroll
    for (int j = 0; j < 6; ++j) s += v[j];
    *sink = s;
}

// roles: one char per wave of the workgroup
__global__ void k(const char* roles, int it_m, int it_v, float* out) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    const char r = roles[wave];
    if (r == 'M') mfma_loop(it_m, &s);
    else if (r == 'V') valu_loop(it_v, &s);
    else if (r == 'I') int_loop(it_v, &s);
    else if (r == 'X') mixed_loop(it_m, &s);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(const char* roles, int it_m, int it_v, float* out, char* droles) {
    const int nw = (int)strlen(roles);
    hipMemcpy(droles, roles, nw, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 64 * nw>>>(droles, 8, 8, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 64 * nw>>>(droles, it_m, it_v, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; char* droles;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&droles, 64);
    const int IM = 20000;       // 16 MFMAs per iteration: 320k MFMAs x 32 cycles = 10.2 M cycles ~ 4.3 ms at 2.4 GHz
    const int IV = 20000;       // 128 fmas per iteration (or 96 integer instr)
    printf("one wave per SIMD (4-wave workgroup, 1 per CU):\n");
    printf("  MMMM  %7.3f ms   (MFMA alone: 16 x %d per wave)\n", run("MMMM", IM, IV, out, droles), IM);
    printf("  VVVV  %7.3f ms   (f32 fma alone: 128 x %d per wave)\n", run("VVVV", IM, IV, out, droles), IV);
    printf("  IIII  %7.3f ms   (integer Philox-like alone: 96 x %d per wave)\n", run("IIII", IM, IV, out, droles), IV);
    printf("  XXXX  %7.3f ms   (one stream: 16 x (MFMA + 6 fma) per iteration, same MFMA count as MMMM)\n", run("XXXX", IM, IV, out, droles));
    printf("two waves per SIMD (8-wave workgroup; waves w and w + 4 share a SIMD):\n");
    printf("  MMMMMMMM  %7.3f ms   (2 MFMA waves per SIMD)\n", run("MMMMMMMM", IM, IV, out, droles));
    printf("  VVVVVVVV  %7.3f ms   (2 fma waves per SIMD)\n", run("VVVVVVVV", IM, IV, out, droles));
    printf("  MMMMVVVV  %7.3f ms   (MFMA wave + fma wave per SIMD: sum or max?)\n", run("MMMMVVVV", IM, IV, out, droles));
    printf("  MMMMIIII  %7.3f ms   (MFMA wave + integer wave per SIMD)\n", run("MMMMIIII", IM, IV, out, droles));
    printf("  VVVVIIII  %7.3f ms   (fma wave + integer wave per SIMD)\n", run("VVVVIIII", IM, IV, out, droles));
    hipFree(out); hipFree(droles);
    return 0;
}
