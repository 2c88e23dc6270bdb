// ccsd_rt.h -- the few runtime hooks the kernel source needs.
//
// Product build (hipcc --offload-arch=gfx950): everything maps onto the HIP runtime and the CDNA4
// intrinsics; this is the only build that ships (libccsd_hip.so).
//
// CCSD_EMU build (g++, tests/emu only): the SAME kernel source is compiled for the host with one
// "thread" per workgroup (threadIdx = 0, blockDim = 1, barriers are no-ops, the MFMA tile
// primitives are plain fmaf loops in the same k order).  It exists so that indexing / weight-layout
// / orchestration bugs are caught on the CPU-only build box before GPU minutes are spent.  It is
// test infrastructure: the python package never loads it and has no CPU fallback.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef CCSD_EMU
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct float2 { float x, y; };
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static dim3 threadIdx(0, 0, 0), blockIdx(0, 0, 0), blockDim(1, 1, 1), gridDim(1, 1, 1);
static inline void __syncthreads() {}
static float* emu_smem = nullptr;
static size_t emu_smem_cap = 0;
#define CCSD_DYN_SMEM(name) float* name = emu_smem
typedef int rtError_t;
#define RT_OK 0
static inline rtError_t rt_malloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 1; }
static inline rtError_t rt_free(void* p) { free(p); return 0; }
static inline rtError_t rt_h2d(void* d, const void* h, size_t n) { memcpy(d, h, n); return 0; }
static inline rtError_t rt_d2d_async(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
static inline rtError_t rt_memset_async(void* d, int v, size_t n, void*) { memset(d, v, n); return 0; }
static inline rtError_t rt_last_error() { return 0; }
static inline const char* rt_error_string(rtError_t) { return "emu"; }
static inline rtError_t rt_set_max_dyn_smem(const void*, size_t) { return 0; }
template <class F>
static inline void emu_launch(dim3 grid, size_t smem, F body) {
    if (smem > emu_smem_cap) {
        free(emu_smem);
        emu_smem = (float*)calloc(smem + 64, 1);
        emu_smem_cap = smem;
    }
    gridDim = grid;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                blockIdx = dim3(x, y, z);
                body();
            }
}
#define CCSD_LAUNCH(kern, grid, block, smem, stream, ...) emu_launch(grid, smem, [&] { (kern)(__VA_ARGS__); })
#define CCSD_NTHREADS 1
#else
#include <hip/hip_runtime.h>
#define CCSD_DYN_SMEM(name) extern __shared__ __align__(16) float name[]
typedef hipError_t rtError_t;
#define RT_OK hipSuccess
static inline rtError_t rt_malloc(void** p, size_t n) { return hipMalloc(p, n ? n : 4); }
static inline rtError_t rt_free(void* p) { return hipFree(p); }
static inline rtError_t rt_h2d(void* d, const void* h, size_t n) { return hipMemcpy(d, h, n, hipMemcpyHostToDevice); }
static inline rtError_t rt_d2d_async(void* d, const void* s, size_t n, void* st) {
    return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st);
}
static inline rtError_t rt_memset_async(void* d, int v, size_t n, void* st) { return hipMemsetAsync(d, v, n, (hipStream_t)st); }
static inline rtError_t rt_last_error() { return hipGetLastError(); }
static inline const char* rt_error_string(rtError_t e) { return hipGetErrorString(e); }
static inline rtError_t rt_set_max_dyn_smem(const void* fn, size_t n) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n);
}
#define CCSD_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#define CCSD_NTHREADS 256
#endif

#define CCSD_DEV __device__ __forceinline__
