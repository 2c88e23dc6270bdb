// What does per-step cross-stream pacing cost on gfx950?  Main stream: K1 (140 us) -> K2 (46 us) -> K3 (5 us), repeated; a side kernel
// Kz (10 us, one wave per CU) is to run beside K2 of every step: [K1][record e][K2][wait f][K3] on the main stream, [wait e][Kz][record f]
// on the side stream.  Compared with the plain chain (no Kz) and with Kz in line on the main stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long cycles, int* out) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    int* d; CK(hipMalloc(&d, 4096));
    hipStream_t s, side; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
    const int steps = 200;
    hipEvent_t e0, e1, ev[2 * steps];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2 * steps; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const double GHz = 2.1;   // (cycle counter of s_memtime runs at ~100 MHz * ..: calibrate by the plain chain below)
    auto us = [&](double u) { return (long long)(u * 100.0); };   // s_memtime-like counter: 100 MHz
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < steps; ++i) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, us(140), d);
                if (mode == 2) CK(hipEventRecord(ev[2 * i], s));
                if (mode == 2) { CK(hipStreamWaitEvent(side, ev[2 * i], 0)); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, side, us(10), d); CK(hipEventRecord(ev[2 * i + 1], side)); }
                hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, us(46), d);
                if (mode == 1) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, us(10), d);
                if (mode == 2) CK(hipStreamWaitEvent(s, ev[2 * i + 1], 0));
                if (mode == 3) { CK(hipEventRecord(ev[2 * i], s)); CK(hipStreamWaitEvent(s, ev[2 * i], 0)); }   // events only, no side work
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, us(5), d);
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const char* nm[] = {"plain chain K1 K2 K3", "Kz in line", "Kz on a side stream, paced by two events per step", "one record + one wait per step, no side work"};
            printf("mode %d (%s): %.2f us per step\n", mode, nm[mode], ms * 1e3 / steps);
        }
    }
    (void)GHz;
    return 0;
}
