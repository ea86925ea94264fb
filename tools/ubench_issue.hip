// Micro-benchmark (not part of the product): how do MFMA 32x32x16 f16 and VALU (softmax-like) work overlap on a
// gfx950 SIMD, within one wave and across co-resident waves?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_issue.hip -o build_abl/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only (14 per iter, 4 accumulators)   MODE 1: VALU only (softmax-like on 32 values)
// MODE 2: MFMA then dependent VALU then dependent MFMA (our tile structure)
// MODE 3: MFMA and VALU independent in one wave (two query blocks, software interleaved by the compiler)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float c) {
    half8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    floatx16 s0, s1, o0, o1, t0, t1;
    for (int r = 0; r < 16; ++r) { s0[r] = threadIdx.x * 1e-3f + r; s1[r] = r * 0.5f; o0[r] = 0; o1[r] = 0; t0[r] = r; t1[r] = -r; }
    float m = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, s1, 0, 0, 0);
            }
        }
        half8 p[4];
        if (MODE == 1 || MODE == 2 || MODE == 3) {
            floatx16& x0 = (MODE == 3) ? t0 : s0;   // MODE 3: VALU works on registers the MFMAs do not touch
            floatx16& x1 = (MODE == 3) ? t1 : s1;
            float mt = x0[0];
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(fmaxf(mt, x0[r]), x1[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            m = fmaxf(m, mt * c);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p0 = __builtin_amdgcn_exp2f(fmaf(x0[r], c, -m));
                float p1 = __builtin_amdgcn_exp2f(fmaf(x1[r], c, -m));
                p[r >> 3][r & 7] = (_Float16)p0;
                p[2 + (r >> 3)][r & 7] = (_Float16)p1;
                if (MODE == 3) { x0[r] = p0 + 1e-3f; x1[r] = p1 - 1e-3f; }
            }
        } else {
            for (int i = 0; i < 4; ++i) p[i] = b;
        }
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, p[kc], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, p[kc], o1, 0, 0, 0);
            }
        } else {
            for (int i = 0; i < 4; ++i) o0[i] += (float)p[i][0];
        }
        if (MODE == 2) { for (int r = 0; r < 16; ++r) { s0[r] = o0[r] * 1e-6f; s1[r] = o1[r] * 1e-6f; } }
    }
    float acc = m;
    for (int r = 0; r < 16; ++r) acc += s0[r] + s1[r] + o0[r] + o1[r] + t0[r] + t1[r];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, float* d) {
    const int iters = 2000;
    for (int bpc = 1; bpc <= 4; ++bpc) {  // blocks of 4 waves per CU -> waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(256), 0, 0, d, 10, 0.1f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 0.1f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // cycles per iteration per SIMD-resident wave set at ~2.1 GHz
        printf("%-28s waves/SIMD=%d: %.1f ns per iter per wave-slot (%.0f cyc @2.1GHz) -> per SIMD %.0f cyc/iter/wave\n", name, bpc,
               1e6 * ms / iters, 2.1e3 * 1e3 * ms / iters, 2.1e6 * ms / iters / bpc);
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 4);
    run<0>("MFMA only (14/iter)", d);
    run<1>("VALU only (softmax 64 val)", d);
    run<2>("MFMA->VALU->MFMA dependent", d);
    run<3>("MFMA + independent VALU", d);
    return 0;
}
