// Microbenchmark (diagnostic, run on the GPU box): issue cost of wave64 VALU instructions on gfx950, in shader cycles per instruction,
// for 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu tools/ubench/valu_rate.hip && /tmp/valu
// The time model of the fp32 kernels here is  sum(MFMA x 32) + sum(VALU x c)  cycles per SIMD: this measures c per instruction class.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k_rate(int iters, float* out, long long* cyc) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {          // v_fma_f32, eight independent chains
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                             "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 1) {   // v_fma_f32, ONE dependent chain
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\t"
                             "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0"
                             : "+v"(a0));
            } else if (MODE == 2) {   // v_mul_lo_u32 (Philox), independent
                asm volatile("v_mul_lo_u32 %0, %0, %0\n\tv_mul_lo_u32 %1, %1, %1\n\tv_mul_lo_u32 %2, %2, %2\n\tv_mul_lo_u32 %3, %3, %3\n\t"
                             "v_mul_lo_u32 %4, %4, %4\n\tv_mul_lo_u32 %5, %5, %5\n\tv_mul_lo_u32 %6, %6, %6\n\tv_mul_lo_u32 %7, %7, %7"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (MODE == 3) {   // v_xor_b32 / v_add_u32 class, independent
                asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3\n\tv_xor_b32 %3, %3, %4\n\t"
                             "v_xor_b32 %4, %4, %5\n\tv_xor_b32 %5, %5, %6\n\tv_xor_b32 %6, %6, %7\n\tv_xor_b32 %7, %7, %0"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (MODE == 4) {   // v_exp_f32 (transcendental), independent
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                             "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 5) {   // v_pk_fma_f32 (two fp32 FMAs per lane and instruction), independent
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
            } else if (MODE == 7) {   // v_mad_u64_u32 (Philox: hi and lo of a 32 x 32 product in one instruction), independent, each with its own carry-out pair
                unsigned long long q0, q1, q2, q3, k0, k1, k2, k3;
                asm volatile("v_mad_u64_u32 %0, %4, %8, %9, 0\n\tv_mad_u64_u32 %1, %5, %9, %10, 0\n\tv_mad_u64_u32 %2, %6, %10, %11, 0\n\tv_mad_u64_u32 %3, %7, %11, %8, 0"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
                u0 ^= (unsigned)q0; u1 ^= (unsigned)(q1 >> 32); u2 ^= (unsigned)q2; u3 ^= (unsigned)(q3 >> 32);
                asm volatile("v_mad_u64_u32 %0, %4, %8, %9, 0\n\tv_mad_u64_u32 %1, %5, %9, %10, 0\n\tv_mad_u64_u32 %2, %6, %10, %11, 0\n\tv_mad_u64_u32 %3, %7, %11, %8, 0"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3) : "v"(u4), "v"(u5), "v"(u6), "v"(u7));
                u4 ^= (unsigned)q0; u5 ^= (unsigned)(q1 >> 32); u6 ^= (unsigned)q2; u7 ^= (unsigned)(q3 >> 32);
            } else {                  // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %0\n\tv_mul_hi_u32 %1, %1, %1\n\tv_mul_hi_u32 %2, %2, %2\n\tv_mul_hi_u32 %3, %3, %3\n\t"
                             "v_mul_hi_u32 %4, %4, %4\n\tv_mul_hi_u32 %5, %5, %5\n\tv_mul_hi_u32 %6, %6, %6\n\tv_mul_hi_u32 %7, %7, %7"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int MODE>
void run(const char* name, float* dout, long long* dc) {
    const int iters = 4000, per = 32;
    printf("%-44s", name);
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipLaunchKernelGGL(k_rate<MODE>, dim3(1), dim3(256 * waves), 0, 0, iters, dout, dc);
        hipDeviceSynchronize();
        long long c[32];
        hipMemcpy(c, dc, 8 * 2 * 4 * waves, hipMemcpyDeviceToHost);
        long long lo = c[0], hi = c[1];
        for (int w = 0; w < 4 * waves; ++w) { if (c[2 * w] < lo) lo = c[2 * w]; if (c[2 * w + 1] > hi) hi = c[2 * w + 1]; }
        printf("  %d wave(s)/SIMD: %.2f (SIMD: %.2f)", waves, (double)(c[1] - c[0]) / (iters * per), (double)(hi - lo) / (iters * per * waves));
    }
    printf("   cycles per instruction of one wave (per instruction issued on the SIMD)\n");
}

int main() {
    float* dout; long long* dc;
    hipMalloc(&dout, 1 << 22); hipMalloc(&dc, 8 * 64);
    run<0>("v_fma_f32 x8 independent", dout, dc);
    run<1>("v_fma_f32 one dependent chain", dout, dc);
    run<2>("v_mul_lo_u32 independent", dout, dc);
    run<6>("v_mul_hi_u32 independent", dout, dc);
    run<3>("v_xor_b32 independent", dout, dc);
    run<4>("v_exp_f32 independent", dout, dc);
    run<5>("v_pk_fma_f32 independent", dout, dc);
    run<7>("v_mad_u64_u32 x8 + 8 v_xor (per 16 instr)", dout, dc);
    return 0;
}
// Measured (profiles/r03_c_valu_rate.txt, cycles per wave64 instruction issued on a SIMD that holds 4 waves): v_fma_f32 2.65, v_xor_b32 2.47,
// v_mul_lo/hi_u32 4.39, v_exp_f32 8.27, v_pk_fma_f32 8.42 (two FMAs: no gain over two v_fma_f32), v_mad_u64_u32 ~13 in this loop; ONE
// wave alone issues a VALU instruction every 5.4-5.8 cycles whether dependent or not, two waves reach 3.4.  (Replacing Philox's
// v_mad_u64_u32 by v_mul_hi_u32 + v_mul_lo_u32 in the kernels was nevertheless slower: k_r2 167.7 -> 172.5 us.)
