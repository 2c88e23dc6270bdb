// Microbenchmark (diagnostic, run on the GPU box): v_mfma_f32_4x4x1_16B_f32 on gfx950 -- operand / result layout and issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/m441 tools/ubench/mfma_4x4x1.hip && /tmp/m441
// 16 independent 4x4 outer products per instruction (k = 1): D_b[i][j] += A_b[i] * B_b[j].  Expected (CDNA3 ISA): lane l supplies
// A_b[i] and B_b[j] with b = l / 4, i = j = l % 4; result register r of lane l holds D_b[r][l % 4].  256 MACs per instruction:
// at the f32 vector rate (32 MACs / cycle / SIMD) that is 8 cycles, i.e. a 4-row strip costs a quarter of a 16x16x4 tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc[r];
}

// mode 0: 4x4x1 independent accumulators; 1: 4x4x1 one dependent chain; 2: 16x16x4 independent; 3: 4x4x1 and v_fma interleaved 1:2
template <int mode>
__global__ void k_rate(int iters, float* out, long long* cyc) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    float v0 = x, v1 = y;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (mode == 0) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, a3, 0, 0, 0);
            } else if (mode == 1) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, a0, 0, 0, 0);
            } else if (mode == 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            } else if (mode == 3) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
                v0 = fmaf(v0, 1.000001f, 1e-7f); v1 = fmaf(v1, 1.000001f, 1e-7f);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0);
                v0 = fmaf(v0, 1.000001f, 1e-7f); v1 = fmaf(v1, 1.000001f, 1e-7f);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a2, 0, 0, 0);
                v0 = fmaf(v0, 1.000001f, 1e-7f); v1 = fmaf(v1, 1.000001f, 1e-7f);
                a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, a3, 0, 0, 0);
                v0 = fmaf(v0, 1.000001f, 1e-7f); v1 = fmaf(v1, 1.000001f, 1e-7f);
            } else {   // 4: two accumulators alternating (each dependent on its own previous result)
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, a1, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {        // per wave: start, end, SIMD id (HW_ID bits 5:4)
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        cyc[3 * (threadIdx.x >> 6) + 0] = t0; cyc[3 * (threadIdx.x >> 6) + 1] = t1; cyc[3 * (threadIdx.x >> 6) + 2] = (hw >> 4) & 3;
    }
}

int main() {
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 100.f + 3.f * l; }
    float *da, *db, *dd, *dout; long long* dc;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024); hipMalloc(&dout, 1 << 22); hipMalloc(&dc, 8 * 3 * 16);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r)
        for (int l = 0; l < 64; ++l) {
            const int blk = l / 4, j = l % 4;
            const float want = a[4 * blk + r] * b[4 * blk + j];          // D_blk[i = r][j] = A_blk[r] * B_blk[j]
            if (d[r * 64 + l] != want) { if (bad < 8) printf("layout mismatch: reg %d lane %d got %g want %g\n", r, l, d[r * 64 + l], want); ++bad; }
        }
    printf("layout (reg r of lane l = D_{l/4}[r][l%%4], A_{l/4}[l%%4], B_{l/4}[l%%4]): %s\n", bad ? "MISMATCH" : "ok");
    const char* names[] = {"4x4x1 x4 independent", "4x4x1 one chain", "16x16x4 x4 independent", "4x4x1 + 2 v_fma each", "4x4x1 two alternating chains"};
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 2000, per = 32;
        for (int waves = 1; waves <= 2; ++waves) {     // waves per SIMD
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(256 * waves), 0, 0, iters, dout, dc); break;
                case 1: hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(256 * waves), 0, 0, iters, dout, dc); break;
                case 2: hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(256 * waves), 0, 0, iters, dout, dc); break;
                case 3: hipLaunchKernelGGL(k_rate<3>, dim3(1), dim3(256 * waves), 0, 0, iters, dout, dc); break;
                default: hipLaunchKernelGGL(k_rate<4>, dim3(1), dim3(256 * waves), 0, 0, iters, dout, dc); break;
            }
            hipDeviceSynchronize();
            long long c[48]; hipMemcpy(c, dc, 8 * 3 * 4 * waves, hipMemcpyDeviceToHost);
            long long lo = c[0], hi = c[1];
            for (int w = 0; w < 4 * waves; ++w) { if (c[3 * w] < lo) lo = c[3 * w]; if (c[3 * w + 1] > hi) hi = c[3 * w + 1]; }
            printf("%-32s %d wave(s)/SIMD: wave 0 %.2f cycles per MFMA; first start -> last end %.2f per MFMA of one wave; SIMD of each wave:", names[mode], waves,
                   (double)(c[1] - c[0]) / (iters * per), (double)(hi - lo) / (iters * per));
            for (int w = 0; w < 4 * waves; ++w) printf(" %lld", c[3 * w + 2]);
            printf("\n");
        }
    }
    return 0;
}
