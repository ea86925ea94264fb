// Wall-clock rate of the fp16 MFMA shapes on the whole chip (two waves per SIMD, independent accumulators, random
// operands unless said otherwise): is the legacy K = 8 form worth using for the last 8 of head dim 40 ?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_shapes.hip -o tools/bin/ubench_mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float scale) {
    half8_t a8, b8;
    half4_t a4, b4;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        a8[e] = (_Float16)(scale * ((int)(s >> 16 & 2047) - 1024) / 512.f);
        s = s * 1664525u + 1013904223u;
        b8[e] = (_Float16)(scale * ((int)(s >> 16 & 2047) - 1024) / 512.f);
    }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    floatx16 c16[4];
    floatx4 c4[4];
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) c16[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) c4[i][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SHAPE == 0) c16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16[i], 0, 0, 0);
            if (SHAPE == 1) c16[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c16[i], 0, 0, 0);
            if (SHAPE == 2) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4[i], 0, 0, 0);
            if (SHAPE == 3) c4[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4[i], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += c16[i][0] + c4[i][0];
    if (t == 12345.f) out[0] = t;
}

template <int SHAPE>
static void run(const char* name, double flop_per_inst, float scale) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000, blocks = 512;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, scale);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double insts_per_simd = 2.0 * iters * 4;  // two waves per SIMD
    printf("%-22s scale %.0f: %.3f ms  -> %.1f ns per instruction and SIMD, %.0f TFLOP/s\n", name, scale, best,
           best * 1e6 / insts_per_simd, (double)blocks * 4 * iters * 4 * flop_per_inst / best / 1e9);
    hipFree(out);
}

int main() {
    for (float scale : {1.f, 0.f}) {  // random operands / zeros
        run<0>("32x32x16_f16", 32768, scale);
        run<1>("32x32x8_f16 (legacy)", 16384, scale);
        run<2>("16x16x32_f16", 16384, scale);
        run<3>("16x16x16_f16 (legacy)", 8192, scale);
    }
    return 0;
}
