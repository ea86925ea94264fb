// Does CODE SIZE set the floor of a small launch?  The same arithmetic (N dependent-free fma per thread) as a rolled loop
// (a few hundred bytes of code) and as straight-line code (8 bytes per v_fma: 1600 -> 12.8 KB, the size of the opt_fast.hip
// kernels), 400 back-to-back launches of 320 x 256 threads, ALTERNATING between two different straight-line kernels so
// that nothing but the instruction cache can keep their code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_icache.hip -o tools/bin/ubench_icache
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N, int SALT>
__global__ void straight(float* p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = p[i], b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {  // fully unrolled: N distinct instructions with distinct constants
        a = fmaf(a, 1.0001f + k * 1e-6f + SALT, 0.5f);
        b = fmaf(b, 1.0002f + k * 1e-6f + SALT, 0.25f);
        c = fmaf(c, 1.0003f + k * 1e-6f + SALT, 0.125f);
        d = fmaf(d, 1.0004f + k * 1e-6f + SALT, 0.0625f);
    }
    p[i] = a + b + c + d;
}
__global__ void rolled(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = p[i], b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll 1
    for (int k = 0; k < n / 4; ++k) {
        a = fmaf(a, 1.0001f, 0.5f);
        b = fmaf(b, 1.0002f, 0.25f);
        c = fmaf(c, 1.0003f, 0.125f);
        d = fmaf(d, 1.0004f, 0.0625f);
    }
    p[i] = a + b + c + d;
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) f(i);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f / 200;
}
int main() {
    float* p;
    (void)hipMalloc(&p, 2048 * 256 * 4);
    (void)hipMemset(p, 0, 2048 * 256 * 4);
    for (int grid : {320, 2048}) {
        printf("grid %d: rolled 400: %.2f us | rolled 1600: %.2f | rolled 6400: %.2f\n", grid,
               timeit([&](int) { hipLaunchKernelGGL(rolled, dim3(grid), dim3(256), 0, 0, p, 400); }),
               timeit([&](int) { hipLaunchKernelGGL(rolled, dim3(grid), dim3(256), 0, 0, p, 1600); }),
               timeit([&](int) { hipLaunchKernelGGL(rolled, dim3(grid), dim3(256), 0, 0, p, 6400); }));
        printf("grid %d: straight-line 400 (3 KB): %.2f us | 1600 (13 KB), same kernel every launch: %.2f | 1600, two kernels alternating: %.2f | "
               "1600, four alternating: %.2f\n", grid,
               timeit([&](int) { hipLaunchKernelGGL((straight<400, 0>), dim3(grid), dim3(256), 0, 0, p); }),
               timeit([&](int) { hipLaunchKernelGGL((straight<1600, 0>), dim3(grid), dim3(256), 0, 0, p); }),
               timeit([&](int i) {
                   if (i & 1) hipLaunchKernelGGL((straight<1600, 0>), dim3(grid), dim3(256), 0, 0, p);
                   else hipLaunchKernelGGL((straight<1600, 1>), dim3(grid), dim3(256), 0, 0, p);
               }),
               timeit([&](int i) {
                   switch (i & 3) {
                       case 0: hipLaunchKernelGGL((straight<1600, 0>), dim3(grid), dim3(256), 0, 0, p); break;
                       case 1: hipLaunchKernelGGL((straight<1600, 1>), dim3(grid), dim3(256), 0, 0, p); break;
                       case 2: hipLaunchKernelGGL((straight<1600, 2>), dim3(grid), dim3(256), 0, 0, p); break;
                       default: hipLaunchKernelGGL((straight<1600, 3>), dim3(grid), dim3(256), 0, 0, p); break;
                   }
               }));
    }
    return 0;
}
