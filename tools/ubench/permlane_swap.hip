// Diagnostic (GPU box): semantics of v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pls tools/ubench/permlane_swap.hip && /tmp/pls
// NOTE (ROCm 7.2 hipcc): __builtin_amdgcn_permlane{16,32}_swap returns both registers, but `r[0] + r[1]` is compiled as
// `r[0] + r[0]` (the two results are coalesced into one register).  The kernels therefore issue the instruction through inline
// asm (ccsd_dev.h: lane_swap16 / lane_swap32); this program checks that form, including the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void lane_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__global__ void k(float* o) {
    float a = threadIdx.x, b = 100.f + threadIdx.x;
    float a16 = a, b16 = b, a32 = a, b32 = b;
    lane_swap16(a16, b16);
    lane_swap32(a32, b32);
    o[threadIdx.x] = a16; o[64 + threadIdx.x] = b16; o[128 + threadIdx.x] = a32; o[192 + threadIdx.x] = b32;
    // all-reduce over the four 16-lane rows
    float x = a, y = a;
    lane_swap16(x, y); x += y; y = x;
    lane_swap32(x, y); x += y;
    o[256 + threadIdx.x] = x;
    // builtin form of the same sum (wrong under ROCm 7.2)
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, a), false, false);
    o[320 + threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
int main() {
    float* d; float h[384];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"swap16: a'", "swap16: b'", "swap32: a'", "swap32: b'"};
    for (int i = 0; i < 4; ++i) {
        printf("%s: rows of 16 lanes start with", nm[i]);
        for (int r = 0; r < 4; ++r) printf(" %g", h[64 * i + 16 * r]);
        printf("   (a = lane, b = 100 + lane)\n");
    }
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[256 + l] != (float)(4 * (l & 15) + 96);
    printf("all-reduce over the rows (asm form): %s (lane 5: %g, want %d)\n", bad ? "WRONG" : "ok", h[256 + 5], 4 * 5 + 96);
    printf("builtin r[0] + r[1] of swap16(a, a), lane 5: %g (a.r0 + a.r1 = %d)\n", h[320 + 5], 5 + 21);
    return 0;
}
