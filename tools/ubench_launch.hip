// Floor of a chain of dependent small launches on one stream: N back-to-back launches of a kernel that reads and writes a
// few bytes per thread (grid x 256 threads), timed with events: what a 4-launch Adam iteration on an 8 x 8 plane cannot
// go below.   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/bin/ubench_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* p, int n, int work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i % n];
    for (int w = 0; w < work; ++w) v = p[(i + (int)v + w * 977) % n] + 1.f;  // dependent L2 round trips
    p[i % n] = v * 0.5f;
}
int main() {
    float* p;
    const int n = 1 << 20;
    (void)hipMalloc(&p, n * 4);
    (void)hipMemset(p, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int grid : {16, 320, 2048})
        for (int work : {0, 2, 6}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, n, work);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("grid %4d x 256 threads, %d dependent loads per thread: %.2f us per launch (400 back to back)\n", grid, work,
                   best * 1e3f / 400);
        }
    return 0;
}
