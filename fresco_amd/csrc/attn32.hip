// fp32 attention for the flow network that produces FRESCO's flows (GMFlow: swin-split self / cross
// attention, global correlation softmax, flow propagation; reference: gmflow/transformer.py:8-17, 92-95,
// gmflow/matching.py:7-36, gmflow/transformer.py:356-372).  The flows feed integer decisions (occlusion
// thresholds, pixel correspondences), and rounding the attention operands to fp16 moves the reference's own
// flows by 0.1 - 2 px, so this path stays in fp32 end to end: v_mfma_f32_32x32x2_f32 for both
// contractions, fp32 online softmax.  Shapes are small (L <= 4096, one head of 128), so the kernel is the
// straightforward form of attn.hip's design, without its DMA / packing machinery:
//
//   * swapped S^T = K Q^T: a lane owns one query (its column of the 32 x 32 C tile) and 16 of a tile's 32
//     keys, lane^32 the other 16 -- row max / sum are in-lane plus one exchange;
//   * the contraction index of either product may be enumerated in any order as long as both operands agree:
//     QK walks d = hi*D/2 + s (each lane reads CONTIGUOUS halves of its K row / Q row), PV walks the keys in
//     the order the accumulator registers hold them, so P is consumed straight from the registers it was
//     computed in (no shuffle, no LDS round trip);
//   * K and V tiles of 32 keys are staged in LDS (rows padded by 4 floats: conflict-free 16-byte reads).
//
// q (B, Lq, D), k (B, Lk, D), v (B, Lk, DV), out (B, Lq, DV), all fp32 row-major.  grid (ceil(Lq/128), B).
#include "common.h"
#include <stdlib.h>

namespace fresco {

// 2^x for x <= 0: the bare v_exp_f32 (1 ulp).  exp2f() wraps it in five more instructions that only matter for results below
// 2^-126 (they come out as 0 here: softmax weights 38 orders of magnitude under the row maximum) -- 85 of the 368 vector
// instructions per tile and wave of attn_f32p_kernel (round 6).
__device__ __forceinline__ float a32_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int D, int DV>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out,
                                                        int Lq, int Lk, int dv_real, float scale_log2,
                                                        const int* __restrict__ only_if) {
    // guarded launches (fresco_attn_f32_guarded): this exact-fp32 form only runs -- and overwrites the split-fp16 result --
    // when the range pass found an operand outside what the fp16 pieces can hold (uniform branch, one scalar load)
    if (only_if && __builtin_amdgcn_readfirstlane(*only_if) == 0) return;
    constexpr int KR = D + 4;    // LDS row strides (floats)
    constexpr int VR = DV + 4;
    constexpr int NDB = DV / 32;  // 32-row blocks of O^T
    __shared__ __attribute__((aligned(16))) float ks[32 * KR];
    __shared__ __attribute__((aligned(16))) float vs[32 * VR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;
    const float* qp = q + ((int64_t)b * Lq + (qrow < Lq ? qrow : Lq - 1)) * D + hi * (D / 2);
    const float* kb = k + (int64_t)b * Lk * D;
    const float* vb = v + (int64_t)b * Lk * dv_real;

    // Q^T fragments: step s contracts d = hi*D/2 + s; the exponent scale is applied in fp32 here
    float qf[D / 2];
#pragma unroll
    for (int s4 = 0; s4 < D / 8; ++s4) {
        const floatx4 t = *reinterpret_cast<const floatx4*>(qp + s4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) qf[s4 * 4 + e] = t[e] * scale_log2;
    }
    floatx16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nT = (Lk + 31) / 32;
    for (int t = 0; t < nT; ++t) {
        __syncthreads();  // the previous tile's fragments have been read
        // stage K (32 x D) and V (32 x DV, columns >= dv_real zero) -- coalesced 16-byte loads where possible
        for (int c = tid; c < 32 * (D / 4); c += 256) {
            const int r = c / (D / 4), d4 = c % (D / 4);
            const int key = t * 32 + r;
            floatx4 val = {0.f, 0.f, 0.f, 0.f};
            if (key < Lk) val = *reinterpret_cast<const floatx4*>(kb + (int64_t)key * D + d4 * 4);
            *reinterpret_cast<floatx4*>(&ks[r * KR + d4 * 4]) = val;
        }
        for (int c = tid; c < 32 * DV; c += 256) {
            const int r = c / DV, d = c % DV;
            const int key = t * 32 + r;
            vs[r * VR + d] = (key < Lk && d < dv_real) ? vb[(int64_t)key * dv_real + d] : 0.f;
        }
        __syncthreads();

        // ---- S^T = K Q^T (exponent arguments: Q carries the scale) --------------------------------------
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kr = &ks[l31 * KR + hi * (D / 2)];
#pragma unroll
        for (int s4 = 0; s4 < D / 8; ++s4) {
            const floatx4 a = *reinterpret_cast<const floatx4*>(kr + s4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[s4 * 4 + e], s, 0, 0, 0);
        }
        // keys beyond Lk (last tile only)
        if ((t + 1) * 32 > Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= Lk) s[r] = -1e30f;
            }
        }
        // ---- online softmax, one query per lane -----------------------------------------------------------
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = a32_exp2(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = a32_exp2(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        // ---- O^T += V^T P^T: step r contracts key (r&3) + 8*(r>>2) + 4*hi -- exactly the key register r holds -
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vr = &vs[((r & 3) + 8 * (r >> 2) + 4 * hi) * VR + l31];
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[db * 32], s[r], o[db], 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < Lq) {
        float* op = out + ((int64_t)b * Lq + qrow) * dv_real;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (d < dv_real) op[d] = o[db][r] * inv;
            }
    }
}


// ------------------------------------------------------------------------------------------------
// Round 4: the same attention on the fp16 matrix pipe at fp32 accuracy.  Every fp32 operand x is split x = xh + xl into two
// halfs (xh = fp16(x), xl = fp16(x - xh): x to a relative 2^-22) and every product a b is taken as ah bh + ah bl + al bh
// (the dropped al bl term is < 2^-22 relative) -- three v_mfma_f32_32x32x16_f16 per 16 contraction steps instead of eight
// v_mfma_f32_32x32x2_f32: 3/16 of the matrix-pipe time, products exact in the fp32 accumulator (the scheme of the Gram /
// S V kernels of opt_fast.hip).  Both contractions: S^T = K Q^T with K, Q split, O^T += V^T P^T with V and the fp32
// probabilities P split; row max / sum / exp2 stay fp32.  The matrix pipe FLUSHES fp16 subnormals, so the lo parts must stay
// normal numbers (> 6.1e-5): every operand is scaled by a power of two before the split -- Q (with the exponent scale
// folded in) and K by 2^6, V by 2^6, P (in [0, 1]) by 2^12 -- and the products are scaled back exactly in fp32 (one
// multiply per score, one on the final normalisation).  Without that, a P below 0.25 loses its lo part: 2e-4 relative on
// the output (measured).  Operands must stay below fp16 range after scaling: |q c|, |k|, |v| < 1000 (GMFlow's
// LayerNorm-bounded tokens and pixel coordinates are).
//   * K tile (32 keys): split while staged, LDS rows [key][D halfs + 16 B pad] for hi and lo;
//   * V tile: split and TRANSPOSED while staged, V^T rows [d][32 key slots + 16 B pad], the slots in the order the S^T
//     accumulator registers hold the keys (slot(key) = (key>>4)*16 + ((key>>2)&1)*8 + ((key>>3)&1)*4 + (key&3)), so the
//     packed P registers ARE the B operand of the second product;
//   * a wave owns 32 queries: Q fragments (hi, lo) resident in registers.
// q (B, Lq, D), k (B, Lk, D), v (B, Lk, dv_real), out (B, Lq, dv_real) fp32.  grid (ceil(Lq/128), B), 256 threads.
// ------------------------------------------------------------------------------------------------
template <int D, int DV, int NPQ>  // NPQ = fp16 pieces of the Q / K operands: 2 (22 bits, 3 MFMAs) or 3 (33 bits, 6 MFMAs)
__global__ __launch_bounds__(256, 2) void attn_f32s_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out,
                                                            int Lq, int Lk, int dv_real, float scale_log2) {
    constexpr int KROW = D * 2 + 16;   // bytes per K row (an odd number of 16-byte units)
    constexpr int VROW = 64 + 16;      // bytes per V^T row (32 key slots)
    constexpr int NDB = DV / 32, NKS = D / 16;
    __shared__ __attribute__((aligned(16))) char kh_s[32 * KROW];
    __shared__ __attribute__((aligned(16))) char kl_s[32 * KROW];
    __shared__ __attribute__((aligned(16))) char km_s[NPQ == 3 ? 32 * KROW : 16];  // (third piece: bits 23 .. 33)
    __shared__ __attribute__((aligned(16))) char vh_s[DV * VROW];
    __shared__ __attribute__((aligned(16))) char vl_s[DV * VROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;
    const float* qp = q + ((int64_t)b * Lq + (qrow < Lq ? qrow : Lq - 1)) * D + hi * 8;
    const float* kb = k + (int64_t)b * Lk * D;
    const float* vb = v + (int64_t)b * Lk * dv_real;

    auto split8 = [](const float (&x)[8], half8_t& h, half8_t& l) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half_t hh = (half_t)x[e];
            h[e] = hh;
            l[e] = (half_t)(x[e] - (float)hh);
        }
    };
    // Q fragments of k-step ks: d = ks*16 + hi*8 .. +7, exponent scale applied in fp32 before the split
    half8_t qh[NKS], ql[NKS], qm[NPQ == 3 ? NKS : 1];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(qp + ks * 16);
        const floatx4 c = *reinterpret_cast<const floatx4*>(qp + ks * 16 + 4);
        const float sq = scale_log2 * 64.f;  // 2^6
        const float x[8] = {a[0] * sq, a[1] * sq, a[2] * sq, a[3] * sq, c[0] * sq, c[1] * sq, c[2] * sq, c[3] * sq};
        split8(x, qh[ks], ql[ks]);
        if (NPQ == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qm[ks][e] = (half_t)(((x[e] - (float)qh[ks][e]) - (float)ql[ks][e]) * 4096.f);
        }
    }
    floatx16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nT = (Lk + 31) / 32;
    for (int t = 0; t < nT; ++t) {
        __syncthreads();  // the previous tile's fragments have been read
        for (int c = tid; c < 32 * (D / 4); c += 256) {  // K: coalesced 16-byte loads, split, 8-byte LDS writes
            const int r = c / (D / 4), d4 = c % (D / 4);
            const int key = t * 32 + r;
            floatx4 val = {0.f, 0.f, 0.f, 0.f};
            if (key < Lk) val = *reinterpret_cast<const floatx4*>(kb + (int64_t)key * D + d4 * 4) * 64.f;  // 2^6
            half4_t h4, l4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half_t hh = (half_t)val[e];
                h4[e] = hh;
                l4[e] = (half_t)(val[e] - (float)hh);
            }
            *reinterpret_cast<half4_t*>(kh_s + r * KROW + d4 * 8) = h4;
            *reinterpret_cast<half4_t*>(kl_s + r * KROW + d4 * 8) = l4;
            if (NPQ == 3) {
                half4_t m4;
#pragma unroll
                for (int e = 0; e < 4; ++e) m4[e] = (half_t)(((val[e] - (float)h4[e]) - (float)l4[e]) * 4096.f);
                *reinterpret_cast<half4_t*>(km_s + r * KROW + d4 * 8) = m4;
            }
        }
        for (int c = tid; c < 32 * DV; c += 256) {  // V: split + transposed into the accumulator's key order
            const int r = c / DV, d = c % DV;
            const int key = t * 32 + r;
            const float val = (key < Lk && d < dv_real) ? vb[(int64_t)key * dv_real + d] * 64.f : 0.f;  // 2^6
            const int pos = (r >> 4) * 16 + ((r >> 2) & 1) * 8 + ((r >> 3) & 1) * 4 + (r & 3);
            const half_t hh = (half_t)val;
            *reinterpret_cast<half_t*>(vh_s + d * VROW + pos * 2) = hh;
            *reinterpret_cast<half_t*>(vl_s + d * VROW + pos * 2) = (half_t)(val - (float)hh);
        }
        __syncthreads();

        // ---- S^T = K Q^T (exponent arguments) ----------------------------------------------------------------------
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if (NPQ == 3) {
            // third pieces xm = x - xh - xl (<= 2^-22 |x|) are stored scaled by 2^12 (they would be fp16 subnormals, which
            // the matrix pipe flushes): their two products go first and are scaled back before the large terms are added
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const half8_t ah = *reinterpret_cast<const half8_t*>(kh_s + l31 * KROW + ks * 32 + hi * 16);
                const half8_t am = *reinterpret_cast<const half8_t*>(km_s + l31 * KROW + ks * 32 + hi * 16);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qm[ks], s, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= 0x1p-12f;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const half8_t ah = *reinterpret_cast<const half8_t*>(kh_s + l31 * KROW + ks * 32 + hi * 16);
            const half8_t al = *reinterpret_cast<const half8_t*>(kl_s + l31 * KROW + ks * 32 + hi * 16);
            if (NPQ == 3) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], s, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= 0x1p-12f;  // undo the 2^6 of Q and of K (exact)
        if ((t + 1) * 32 > Lk) {  // keys beyond Lk (last tile only)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= Lk) s[r] = -1e30f;
            }
        }
        // ---- online softmax, one query per lane (fp32) -------------------------------------------------------------
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = a32_exp2(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = a32_exp2(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        // ---- O^T += V^T P^T: k-step st contracts the 16 keys whose probabilities registers 8 st .. 8 st + 7 hold --------
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = s[st * 8 + e] * 4096.f;  // 2^12
            half8_t ph, pl;
            split8(x, ph, pl);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const half8_t vh8 = *reinterpret_cast<const half8_t*>(vh_s + (db * 32 + l31) * VROW + st * 32 + hi * 16);
                const half8_t vl8 = *reinterpret_cast<const half8_t*>(vl_s + (db * 32 + l31) * VROW + st * 32 + hi * 16);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph, o[db], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 0x1p-18f / l_tot;  // (the accumulators carry the 2^12 of P and the 2^6 of V)
    if (qrow < Lq) {
        float* op = out + ((int64_t)b * Lq + qrow) * dv_real;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (d < dv_real) op[d] = o[db][r] * inv;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4, second step: K / V are split ONCE per launch, not once per 128-query workgroup.  attn_f32s_kernel above
// converts every 32-key tile in every workgroup that reads it (a window of 1024 tokens: 8 times), and its matrix pipe
// idles while all 256 threads run the conversion: 102 TFLOP/s at (B 64, L 1024, D 128).  Here
//   kv_split_kernel   writes, per (batch entry, 32-key tile), the exact LDS image the attention loop consumes -- K hi / lo /
//                     third piece as [key][D halfs + 16 B] rows, V^T hi / lo as [d][32 key slots + 16 B] rows in the
//                     accumulator's key order, all scaled as above -- padded to whole 4 KiB so that every wave copies the
//                     same number of 1 KiB pieces;
//   attn_f32p_kernel  copies the images by LDS-DMA into ONE K buffer and ONE V buffer and alternates them: while the waves
//                     multiply K(t) Q^T the V^T(t) image lands, while they multiply V^T(t) P^T the K(t+1) image lands
//                     (four barriers per tile, counted vmcnt); no conversion, no staging registers, no ds_write.
// Same arithmetic as attn_f32s_kernel<D, DV, 3>, bit for bit (the pieces are the same numbers).
// ------------------------------------------------------------------------------------------------
template <int D, int DV>
struct A32Img {
    static constexpr int KROW = D * 2 + 16, VROW = 64 + 16;
    static constexpr int KIMG = 3 * 32 * KROW, VIMG = 2 * DV * VROW;
    static constexpr int NKP = (KIMG + 4095) / 4096, NVP = (VIMG + 4095) / 4096;  // 1 KiB pieces PER WAVE (4 waves)
    static constexpr int KIMGP = NKP * 4096, VIMGP = NVP * 4096;
    static constexpr int TILE = KIMGP + VIMGP;
};

template <int D, int DV>
__global__ __launch_bounds__(256) void kv_split_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                        char* __restrict__ img, int Lk, int dv_real, int* __restrict__ flag) {
    // flag != NULL (fresco_attn_f32_guarded_ws): the range test of the guarded entry on the values this pass reads anyway --
    // |k|, |v| >= 1000 (or NaN) would overflow the hi piece after the 2^6 pre-scale: OR 1 into the launch's flag word
    using I = A32Img<D, DV>;
    constexpr int KROW = I::KROW, VROW = I::VROW;
    __shared__ __attribute__((aligned(16))) char s[I::TILE];
    const int t = blockIdx.x, b = blockIdx.y, nT = gridDim.x, tid = threadIdx.x;
    const float* kb = k + (int64_t)b * Lk * D;
    const float* vb = v + (int64_t)b * Lk * dv_real;
    for (int i = tid; i < I::TILE / 16; i += 256) reinterpret_cast<uint4*>(s)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    char* kh_s = s;
    char* kl_s = s + 32 * KROW;
    char* km_s = s + 2 * 32 * KROW;
    char* vh_s = s + I::KIMGP;
    char* vl_s = vh_s + DV * VROW;
    bool bad = false;
    for (int c = tid; c < 32 * (D / 4); c += 256) {
        const int r = c / (D / 4), d4 = c % (D / 4);
        const int key = t * 32 + r;
        floatx4 val = {0.f, 0.f, 0.f, 0.f};
        if (key < Lk) {
            const floatx4 raw = *reinterpret_cast<const floatx4*>(kb + (int64_t)key * D + d4 * 4);
            bad |= !(fabsf(raw[0]) < 1000.f) || !(fabsf(raw[1]) < 1000.f) || !(fabsf(raw[2]) < 1000.f) || !(fabsf(raw[3]) < 1000.f);
            val = raw * 64.f;
        }
        half4_t h4, l4, m4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const half_t hh = (half_t)val[e];
            const half_t ll = (half_t)(val[e] - (float)hh);
            h4[e] = hh;
            l4[e] = ll;
            m4[e] = (half_t)(((val[e] - (float)hh) - (float)ll) * 4096.f);
        }
        *reinterpret_cast<half4_t*>(kh_s + r * KROW + d4 * 8) = h4;
        *reinterpret_cast<half4_t*>(kl_s + r * KROW + d4 * 8) = l4;
        *reinterpret_cast<half4_t*>(km_s + r * KROW + d4 * 8) = m4;
    }
    for (int c = tid; c < 32 * DV; c += 256) {
        const int r = c / DV, d = c % DV;
        const int key = t * 32 + r;
        const float vraw = (key < Lk && d < dv_real) ? vb[(int64_t)key * dv_real + d] : 0.f;
        bad |= !(fabsf(vraw) < 1000.f);
        const float val = vraw * 64.f;
        const int pos = (r >> 4) * 16 + ((r >> 2) & 1) * 8 + ((r >> 3) & 1) * 4 + (r & 3);
        const half_t hh = (half_t)val;
        *reinterpret_cast<half_t*>(vh_s + d * VROW + pos * 2) = hh;
        *reinterpret_cast<half_t*>(vl_s + d * VROW + pos * 2) = (half_t)(val - (float)hh);
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(img + ((int64_t)b * nT + t) * I::TILE);
    for (int i = tid; i < I::TILE / 16; i += 256) dst[i] = reinterpret_cast<const uint4*>(s)[i];
    if (flag && __any(bad) && (tid & 63) == 0) atomicOr(flag, 1);
}

template <int N_>
__device__ __forceinline__ void a32_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N_) : "memory");
}

// Round 6: TWO tile buffers and NW = 4 or 8 waves (128 or 256 queries) per workgroup.  The single-buffer form of rounds 4-5
// could only request K(t + 1) once every wave was done with K(t), i.e. half a tile ahead: with the image coming from HBM /
// MALL (a 1024-token window has 8 query blocks to share a tile, and they run in lockstep: the first to ask pays the miss and
// the others wait on it) that latency sat on every tile -- 5.6 us per tile at (B 64, L 1024) against 3.1 at (B 16, L 4096),
// where 32 query blocks share and somebody has usually been there before.  Now tile t + 1 is requested as a whole (K | V,
// one contiguous image) at the top of tile t into the other buffer: a full tile of lead, ONE barrier per tile instead of
// four, no counted waits (what is outstanding at the top of tile t is tile t, requested a tile ago).  With 8 waves one
// image stream feeds 256 queries: half the LDS-DMA bytes per flop.  Same arithmetic on the same pieces: identical bits.
// Worth 4 % (197.8 -> 188.9 us for range pass + split + attention at (B 64, L 1024), 420 -> 413 at (B 16, L 4096)): the
// counters of this form (tools/pmc_attn32.sh, profiles/r06_pmc_attn32_B64_L1024.csv) say what bounds it instead -- 4.72 M
// MFMAs x 32 cycles = 147 k busy cycles per SIMD of 305 k (48 %), 24.1 M VALU instructions = 23.5 k per SIMD ~ 105 k port
// cycles + 32 k of MFMA issue: as in the flash kernel the vector strand (368 instructions per tile and wave: P's (hi, lo)
// split, the rescale of O, exp2, the two 2^-12 scalings) and the matrix strand of a SIMD add up instead of overlapping;
// LDS bank conflicts 0, L2 hit rate 83 %.
template <int D, int DV, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_f32p_kernel(const float* __restrict__ q,
                                                                              const char* __restrict__ img,
                                                                              float* __restrict__ out, int Lq, int Lk,
                                                                              int dv_real, float scale_log2,
                                                                              int* __restrict__ flag) {
    using I = A32Img<D, DV>;
    constexpr int KROW = I::KROW, VROW = I::VROW;
    constexpr int NDB = DV / 32, NKS = D / 16;
    constexpr int PPW = I::TILE / 1024 / NW;  // 1 KiB pieces per wave and tile
    static_assert(I::TILE % (1024 * NW) == 0, "every wave copies the same number of pieces");
    extern __shared__ __attribute__((aligned(16))) char tbuf[];  // [2][TILE]: K image | V image of a tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // workgroup -> (problem b, query block qb), query blocks of a problem on consecutive ids.  (Round 6, measured and not
    // kept: an XCD-aware order that keeps a problem's query blocks on ONE XCD's L2 -- 197.8 / 204.5 us against 202.7 / 210.1
    // at (B 64, L 1024), profiles/r06_ab_attn32_forms.txt: the launch is not bound by where its images come from, see the
    // counters quoted below.)
    const int nqb = (Lq + NW * 32 - 1) / (NW * 32);
    const int b = blockIdx.x / nqb, qb = blockIdx.x - b * nqb;
    const int qrow = qb * (NW * 32) + wave * 32 + l31;
    const float* qp = q + ((int64_t)b * Lq + (qrow < Lq ? qrow : Lq - 1)) * D + hi * 8;
    const int nT = (Lk + 31) / 32;
    const char* ib = img + (int64_t)b * nT * I::TILE;
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)tbuf);
    const uint32_t voff = (uint32_t)lane * 16;
    auto stage = [&](int t) __attribute__((always_inline)) {  // wave w copies KiB w, w + NW, ... of tile t's image
        const char* src = ib + (int64_t)t * I::TILE + wave * 1024;
        const uint32_t dst = lds0 + (uint32_t)((t & 1) * I::TILE + wave * 1024);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src + i * (NW * 1024)),
                         "s"(dst + (uint32_t)(i * (NW * 1024)))
                         : "memory");
    };
    stage(0);

    half8_t qh[NKS], ql[NKS], qm[NKS];
    bool qbad = false;
    const float qlim = 1000.f / scale_log2;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(qp + ks * 16);
        const floatx4 c = *reinterpret_cast<const floatx4*>(qp + ks * 16 + 4);
        const float sq = scale_log2 * 64.f;
        // the guarded entry's test |q scale log2 e| < 1000, on the RAW values: a second use of the products below would change
        // which of them hipcc contracts into FMAs, and with it the last bit of the pieces (attn_f32s_kernel must get the same)
#pragma unroll
        for (int e = 0; e < 4; ++e) qbad |= !(fabsf(a[e]) < qlim) || !(fabsf(c[e]) < qlim);
        const float x[8] = {a[0] * sq, a[1] * sq, a[2] * sq, a[3] * sq, c[0] * sq, c[1] * sq, c[2] * sq, c[3] * sq};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half_t hh = (half_t)x[e];
            const half_t ll = (half_t)(x[e] - (float)hh);
            qh[ks][e] = hh;
            ql[ks][e] = ll;
            qm[ks][e] = (half_t)(((x[e] - (float)hh) - (float)ll) * 4096.f);
        }
    }
    if (flag && __any(qbad) && lane == 0) atomicOr(flag, 1);
    floatx16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    for (int t = 0; t < nT; ++t) {
        // tile t has landed for everyone, and everyone is done with tile t - 1: its buffer takes tile t + 1
        a32_wait_barrier<0>();
        if (t + 1 < nT) stage(t + 1);
        const char* kh_s = tbuf + (t & 1) * I::TILE;
        const char* kl_s = kh_s + 32 * KROW;
        const char* km_s = kh_s + 2 * 32 * KROW;
        const char* vh_s = kh_s + I::KIMGP;
        const char* vl_s = vh_s + DV * VROW;
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const half8_t ah = *reinterpret_cast<const half8_t*>(kh_s + l31 * KROW + ks * 32 + hi * 16);
            const half8_t am = *reinterpret_cast<const half8_t*>(km_s + l31 * KROW + ks * 32 + hi * 16);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, qh[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qm[ks], s, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= 0x1p-12f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const half8_t ah = *reinterpret_cast<const half8_t*>(kh_s + l31 * KROW + ks * 32 + hi * 16);
            const half8_t al = *reinterpret_cast<const half8_t*>(kl_s + l31 * KROW + ks * 32 + hi * 16);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], s, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] *= 0x1p-12f;  // undo the 2^6 of Q and of K (exact)
        if ((t + 1) * 32 > Lk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= Lk) s[r] = -1e30f;
            }
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = a32_exp2(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = a32_exp2(s[r] - m_new);
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            half8_t ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = s[st * 8 + e] * 4096.f;
                const half_t hh = (half_t)x;
                ph[e] = hh;
                pl[e] = (half_t)(x - (float)hh);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const half8_t vh8 = *reinterpret_cast<const half8_t*>(vh_s + (db * 32 + l31) * VROW + st * 32 + hi * 16);
                const half8_t vl8 = *reinterpret_cast<const half8_t*>(vl_s + (db * 32 + l31) * VROW + st * 32 + hi * 16);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, ph, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh8, pl, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl8, ph, o[db], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 0x1p-18f / l_tot;
    if (qrow < Lq) {
        float* op = out + ((int64_t)b * Lq + qrow) * dv_real;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (d < dv_real) op[d] = o[db][r] * inv;
            }
    }
}

template <int D, int DV>
static int launch_attn32p(const float* q, const float* k, const float* v, float* out, char* img, int B, int Lq, int Lk, int dv,
                          float scale, hipStream_t st, int* flag = nullptr) {
    ProfScope ps(FRESCO_PROF_ATTN_F32, B, Lq, Lk, D, st);
    const int nT = (Lk + 31) / 32;
    hipLaunchKernelGGL((kv_split_kernel<D, DV>), dim3(nT, B), dim3(256), 0, st, k, v, img, Lk, dv, flag);
    using I = A32Img<D, DV>;
    const int lds = 2 * I::TILE;
    // 256-query workgroups (8 waves: one image stream per 256 queries) when they alone fill the chip and the tile's pieces
    // divide by 8 waves; else 128-query workgroups
    constexpr bool can8 = I::TILE % 8192 == 0;
    const int64_t wg8 = (int64_t)((Lq + 255) / 256) * B;
    if (can8 && wg8 >= 256) {
        if constexpr (can8) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32p_kernel<D, DV, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((attn_f32p_kernel<D, DV, 8>), dim3((unsigned)wg8), dim3(512), lds, st, q, img, out, Lq, Lk, dv,
                               scale * 1.4426950408889634f, flag);
        }
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32p_kernel<D, DV, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((attn_f32p_kernel<D, DV, 4>), dim3(((Lq + 127) / 128) * B), dim3(256), lds, st, q, img, out, Lq, Lk,
                           dv, scale * 1.4426950408889634f, flag);
    }
    return check_launch();
}

template <int D, int DV>
static int launch_attn32(const float* q, const float* k, const float* v, float* out, int B, int Lq, int Lk, int dv,
                         float scale, hipStream_t st) {
    ProfScope ps(FRESCO_PROF_ATTN_F32, B, Lq, Lk, D, st);
    // FRESCO_ATTN_F32=mfma32: the fp32-MFMA kernel (exact fp32 products; A/B measurements, and operands beyond fp16 range)
    static const int use_f32_mfma = [] {
        const char* e = getenv("FRESCO_ATTN_F32");
        return (e && e[0] == 'm') ? 1 : ((e && e[0] == '2') ? 2 : 0);
    }();
    if (use_f32_mfma == 2)  // (22-bit logits: 3 MFMAs per product in both contractions; measurements)
        hipLaunchKernelGGL((attn_f32s_kernel<D, DV, 2>), dim3((Lq + 127) / 128, B), dim3(256), 0, st, q, k, v, out, Lq, Lk,
                           dv, scale * 1.4426950408889634f);
    else if (use_f32_mfma)
        hipLaunchKernelGGL((attn_f32_kernel<D, DV>), dim3((Lq + 127) / 128, B), dim3(256), 0, st, q, k, v, out, Lq, Lk,
                           dv, scale * 1.4426950408889634f, (const int*)nullptr);
    else
        hipLaunchKernelGGL((attn_f32s_kernel<D, DV, 3>), dim3((Lq + 127) / 128, B), dim3(256), 0, st, q, k, v, out, Lq, Lk,
                           dv, scale * 1.4426950408889634f);
    return check_launch();
}

// Range pass of the guarded entry: the split-fp16 kernels scale every operand by 2^6 before the split (see above), so
// |q * scale * log2(e)|, |k|, |v| must stay below ~1000 or the hi piece overflows fp16 (inf -> NaN out of the MFMA).
// One grid-stride pass over q, k, v (a few MB in the flow network) raises *flag when any element is beyond its limit
// or not finite; the exact-fp32 kernel behind it then recomputes the launch.
__global__ __launch_bounds__(256) void a32_range_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int64_t nq, int64_t nk, int64_t nv,
                                                         float lim_q, float lim_kv, int* __restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) bad |= !(fabsf(q[i]) < lim_q);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += stride) bad |= !(fabsf(k[i]) < lim_kv);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) bad |= !(fabsf(v[i]) < lim_kv);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <int D, int DV>
static int launch_attn32_fallback(const float* q, const float* k, const float* v, float* out, int B, int Lq, int Lk, int dv,
                                  float scale, const int* flag, hipStream_t st) {
    hipLaunchKernelGGL((attn_f32_kernel<D, DV>), dim3((Lq + 127) / 128, B), dim3(256), 0, st, q, k, v, out, Lq, Lk, dv,
                       scale * 1.4426950408889634f, flag);
    return check_launch();
}

}  // namespace fresco

using namespace fresco;

extern "C" int fresco_attn_f32(const float* q, const float* k, const float* v, float* out, int B, int Lq, int Lk,
                               int D, int Dv, float scale, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || Lq <= 0 || Lk <= 0 || D <= 0 || Dv <= 0 || !(scale > 0.f))
        return FRESCO_EINVAL;
    if (B > 65535) return FRESCO_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
#define FRESCO_A32(DD)                                                                           \
    if (D == DD) {                                                                               \
        if (Dv <= 32) return launch_attn32<DD, 32>(q, k, v, out, B, Lq, Lk, Dv, scale, st);      \
        if (Dv <= 64) return launch_attn32<DD, 64>(q, k, v, out, B, Lq, Lk, Dv, scale, st);      \
        if (Dv <= 128) return launch_attn32<DD, 128>(q, k, v, out, B, Lq, Lk, Dv, scale, st);    \
        return FRESCO_EUNSUPPORTED;                                                              \
    }
    FRESCO_A32(32)
    FRESCO_A32(64)
    FRESCO_A32(128)
#undef FRESCO_A32
    return FRESCO_EUNSUPPORTED;
}

// ---- the same with a caller-provided workspace: K / V are split once per launch (kv_split_kernel + attn_f32p_kernel) ----
template <int D, int DV>
static size_t a32_ws_bytes(int B, int Lk) {
    return (size_t)B * ((Lk + 31) / 32) * A32Img<D, DV>::TILE;
}

extern "C" size_t fresco_attn_f32_workspace_bytes(int B, int Lk, int D, int Dv) {
    if (B <= 0 || Lk <= 0 || Dv <= 0 || Dv > 128) return 0;
#define FRESCO_A32W(DD)                                                  \
    if (D == DD) {                                                       \
        if (Dv <= 32) return a32_ws_bytes<DD, 32>(B, Lk);                \
        if (Dv <= 64) return a32_ws_bytes<DD, 64>(B, Lk);                \
        return a32_ws_bytes<DD, 128>(B, Lk);                             \
    }
    FRESCO_A32W(32)
    FRESCO_A32W(64)
    FRESCO_A32W(128)
#undef FRESCO_A32W
    return 0;
}

extern "C" int fresco_attn_f32_ws(const float* q, const float* k, const float* v, float* out, void* workspace,
                                  size_t workspace_bytes, int B, int Lq, int Lk, int D, int Dv, float scale, void* stream) {
    if (!q || !k || !v || !out || !workspace || B <= 0 || Lq <= 0 || Lk <= 0 || D <= 0 || Dv <= 0 || !(scale > 0.f))
        return FRESCO_EINVAL;
    if (B > 65535) return FRESCO_EUNSUPPORTED;
    const size_t need = fresco_attn_f32_workspace_bytes(B, Lk, D, Dv);
    if (need == 0) return FRESCO_EUNSUPPORTED;
    if (workspace_bytes < need) return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    char* img = static_cast<char*>(workspace);
#define FRESCO_A32P(DD)                                                                                  \
    if (D == DD) {                                                                                       \
        if (Dv <= 32) return launch_attn32p<DD, 32>(q, k, v, out, img, B, Lq, Lk, Dv, scale, st);        \
        if (Dv <= 64) return launch_attn32p<DD, 64>(q, k, v, out, img, B, Lq, Lk, Dv, scale, st);        \
        return launch_attn32p<DD, 128>(q, k, v, out, img, B, Lq, Lk, Dv, scale, st);                     \
    }
    FRESCO_A32P(32)
    FRESCO_A32P(64)
    FRESCO_A32P(128)
#undef FRESCO_A32P
    return FRESCO_EUNSUPPORTED;
}

// ---- guarded form: no range limit on the operands.  `flag` = one int32 of device memory owned by the caller
// for the duration of the call.  Range pass -> split-fp16 kernels (with the workspace when one is given, else the
// per-workgroup staging form) -> the exact-fp32 MFMA kernel, which returns at once unless the range pass raised the flag.
// In range (every call of the flow network): the results of fresco_attn_f32 / _ws bit for bit, + ~2 short launches. ----
extern "C" int fresco_attn_f32_guarded(const float* q, const float* k, const float* v, float* out, void* workspace,
                                       size_t workspace_bytes, int* flag, int B, int Lq, int Lk, int D, int Dv, float scale,
                                       void* stream) {
    if (!flag) return FRESCO_EINVAL;
    if (!q || !k || !v || !out || B <= 0 || Lq <= 0 || Lk <= 0 || D <= 0 || Dv <= 0 || !(scale > 0.f)) return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return FRESCO_ELAUNCH;
    const int64_t nq = (int64_t)B * Lq * D, nk = (int64_t)B * Lk * D, nv = (int64_t)B * Lk * Dv;
    const int64_t nmax = nq > nk ? (nq > nv ? nq : nv) : (nk > nv ? nk : nv);
    int blocks = (int)((nmax + 256 * 8 - 1) / (256 * 8));
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    const float lim = 1000.f;
    hipLaunchKernelGGL(a32_range_kernel, dim3(blocks), dim3(256), 0, st, q, k, v, nq, nk, nv,
                       lim / (scale * 1.4426950408889634f), lim, flag);
    int rc = workspace ? fresco_attn_f32_ws(q, k, v, out, workspace, workspace_bytes, B, Lq, Lk, D, Dv, scale, stream)
                       : fresco_attn_f32(q, k, v, out, B, Lq, Lk, D, Dv, scale, stream);
    if (rc != FRESCO_OK) return rc;
#define FRESCO_A32G(DD)                                                                                     \
    if (D == DD) {                                                                                          \
        if (Dv <= 32) return launch_attn32_fallback<DD, 32>(q, k, v, out, B, Lq, Lk, Dv, scale, flag, st);  \
        if (Dv <= 64) return launch_attn32_fallback<DD, 64>(q, k, v, out, B, Lq, Lk, Dv, scale, flag, st);  \
        return launch_attn32_fallback<DD, 128>(q, k, v, out, B, Lq, Lk, Dv, scale, flag, st);               \
    }
    FRESCO_A32G(32)
    FRESCO_A32G(64)
    FRESCO_A32G(128)
#undef FRESCO_A32G
    return FRESCO_EUNSUPPORTED;
}

// ---- guarded workspace form WITHOUT the range pass and the flag memset (round 6: 20 us of short launches around every one of
// the flow network's 26 attention calls).  `zero_flag` = one int32 of device memory that the CALLER guarantees to be zero in
// stream order when the call is issued and that nobody else touches until the call's kernels are done (fresco_amd.ops hands out
// the words of a zero-filled pool, each once).  The k / v range test runs inside the split pass (which reads them anyway), the
// q test in the attention kernel's Q prologue; the exact-fp32 kernel behind them returns at once unless one of them raised
// the flag.  Same results as fresco_attn_f32_guarded with a workspace, bit for bit, in range and out of range.
extern "C" int fresco_attn_f32_guarded_ws(const float* q, const float* k, const float* v, float* out, void* workspace,
                                          size_t workspace_bytes, int* zero_flag, int B, int Lq, int Lk, int D, int Dv,
                                          float scale, void* stream) {
    if (!zero_flag || !workspace) return FRESCO_EINVAL;
    if (!q || !k || !v || !out || B <= 0 || Lq <= 0 || Lk <= 0 || D <= 0 || Dv <= 0 || !(scale > 0.f)) return FRESCO_EINVAL;
    if (B > 65535) return FRESCO_EUNSUPPORTED;
    const size_t need = fresco_attn_f32_workspace_bytes(B, Lk, D, Dv);
    if (need == 0) return FRESCO_EUNSUPPORTED;
    if (workspace_bytes < need) return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    char* img = static_cast<char*>(workspace);
#define FRESCO_A32GW(DD)                                                                                                  \
    if (D == DD) {                                                                                                        \
        int rc;                                                                                                           \
        if (Dv <= 32) {                                                                                                   \
            rc = launch_attn32p<DD, 32>(q, k, v, out, img, B, Lq, Lk, Dv, scale, st, zero_flag);                          \
            return rc != FRESCO_OK ? rc : launch_attn32_fallback<DD, 32>(q, k, v, out, B, Lq, Lk, Dv, scale, zero_flag, st);   \
        }                                                                                                                 \
        if (Dv <= 64) {                                                                                                   \
            rc = launch_attn32p<DD, 64>(q, k, v, out, img, B, Lq, Lk, Dv, scale, st, zero_flag);                          \
            return rc != FRESCO_OK ? rc : launch_attn32_fallback<DD, 64>(q, k, v, out, B, Lq, Lk, Dv, scale, zero_flag, st);   \
        }                                                                                                                 \
        rc = launch_attn32p<DD, 128>(q, k, v, out, img, B, Lq, Lk, Dv, scale, st, zero_flag);                             \
        return rc != FRESCO_OK ? rc : launch_attn32_fallback<DD, 128>(q, k, v, out, B, Lq, Lk, Dv, scale, zero_flag, st);      \
    }
    FRESCO_A32GW(32)
    FRESCO_A32GW(64)
    FRESCO_A32GW(128)
#undef FRESCO_A32GW
    return FRESCO_EUNSUPPORTED;
}
