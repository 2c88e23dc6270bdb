// Does hipExtAnyOrderLaunch let a kernel overlap its predecessor in the SAME stream on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Two spin kernels of T us each on 256 single-wave workgroups: back to back they take 2T; overlapped T.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(long long cycles, int* out) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
    int* d; hipMalloc(&d, 4096);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long cyc = 200000;   // ~100 us
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, cyc, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, cyc, d);
            else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
            else { hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
                   hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, cyc, d); }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "plain, plain" : mode == 1 ? "plain, any-order" : "plain, any-order, plain", ms * 1e3);
        }
    }
    return 0;
}
