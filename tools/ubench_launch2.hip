// Floor of a chain of dependent launches that PASS DATA through memory: launch i reads what launch i-1 wrote (ping-pong
// between two buffers of `mb` megabytes, every thread 16-byte loads / stores, grid sized to the data), 200 back to back.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch2.hip -o tools/bin/ubench_launch2
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float4* __restrict__ in, float4* __restrict__ out, int n4, int per) {
    const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * per;
    for (int j = 0; j < per; ++j) {
        const int i = i0 + j;
        if (i < n4) {
            float4 v = in[(i + n4 / 2 + 64) % n4];  // (coalesced; another workgroup's, usually another XCD's, output)
            v.x += 1.f;
            out[i] = v;
        }
    }
}
int main() {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mb : {1, 4, 16, 64}) {
        const int n4 = mb * (1 << 20) / 16;
        float4 *a, *b;
        (void)hipMalloc(&a, (size_t)n4 * 16);
        (void)hipMalloc(&b, (size_t)n4 * 16);
        (void)hipMemset(a, 0, (size_t)n4 * 16);
        (void)hipMemset(b, 0, (size_t)n4 * 16);
        for (int per : {1, 4}) {
            const int grid = (n4 / per + 255) / 256;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                for (int i = 0; i < 200; ++i) {
                    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b, n4, per);
                }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%3d MB per launch, grid %6d x 256, %d x 16 B per thread: %.2f us per launch = %.2f TB/s (read + write)\n", mb, grid, per,
                   best * 1e3f / 200, 2.0 * mb * 1.048576 / (best * 1e3 / 200) );
        }
        (void)hipFree(a);
        (void)hipFree(b);
    }
    return 0;
}
