// Sustained MFMA rate of the whole chip in wall-clock terms (not cycles): every wave issues independent
// v_mfma_f32_32x32x16_f16 back to back.  Prints TFLOP/s for 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_peak.hip -o build_abl/ubench_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void peak_kernel(float* out, int iters) {
    floatx16 acc[NACC];
    half8_t a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(0.001f * (threadIdx.x + e));
        b[e] = (_Float16)(0.002f * (threadIdx.x - e));
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;  // 4 waves per block: one per SIMD
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((peak_kernel<4>), dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 4 * 32768.0;
            if (rep == 2) printf("waves/SIMD %d: %.3f ms, %.0f TFLOP/s\n", wps, ms, flop / ms / 1e9);
        }
    }
    // long run: does the rate hold for ~100 ms?
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((peak_kernel<4>), dim3(1024), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("20 launches back to back: %.1f ms, %.0f TFLOP/s\n", ms, 20.0 * 1024 * 4 * iters * 4 * 32768.0 / ms / 1e9);
    return 0;
}
