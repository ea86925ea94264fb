// Wall-clock rate of the int8 MFMA shapes next to v_mfma_f32_32x32x16_f16 on the whole chip (two waves per SIMD, four
// independent accumulators, random operands): what an int8 form of the S V product (S in {-1, 0, 1} is exact in int8,
// V as three signed 8-bit digits) could count on.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_i8.hip -o tools/bin/ubench_mfma_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int mode) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    half8_t a8, b8;
    intx4 ai, bi;
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        a8[e] = mode ? (_Float16)(((int)(s >> 16 & 2047) - 1024) / 512.f) : (_Float16)0.f;
        s = s * 1664525u + 1013904223u;
        b8[e] = mode ? (_Float16)(((int)(s >> 16 & 2047) - 1024) / 512.f) : (_Float16)0.f;
    }
    for (int e = 0; e < 4; ++e) {
        s = s * 1664525u + 1013904223u;
        ai[e] = mode ? (int)s : 0;
        s = s * 1664525u + 1013904223u;
        // mode 2: the B operand holds only -1 / 0 / +1 bytes (a sign matrix)
        bi[e] = mode == 2 ? (int)((s & 0x01010101u) * ((s >> 8 & 1) ? 0xffu : 1u)) : (mode ? (int)s : 0);
    }
    floatx16 cf[4];
    intx16 ci[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) {
            cf[i][r] = 0.f;
            ci[i][r] = 0;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SHAPE == 0) cf[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, cf[i], 0, 0, 0);
            if (SHAPE == 1) ci[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, ci[i], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) t += cf[i][0] + (float)ci[i][0];
    if (t == 12345.f) out[0] = t;
}

template <int SHAPE>
static void run(const char* name, double op_per_inst, int mode) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000, blocks = 512;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double insts_per_simd = 2.0 * iters * 4;  // two waves per SIMD
    printf("%-22s %-14s: %.3f ms  -> %.1f ns per instruction and SIMD, %.0f Top/s\n", name,
           mode == 0 ? "zeros" : (mode == 1 ? "random" : "random x signs"), best, best * 1e6 / insts_per_simd,
           (double)blocks * 4 * iters * 4 * op_per_inst / best / 1e9);
    hipFree(out);
}

int main() {
    for (int mode : {1, 2, 0}) {
        if (mode != 2) run<0>("f32_32x32x16_f16", 32768, mode);
        run<1>("i32_32x32x32_i8", 65536, mode);
    }
    return 0;
}
