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

namespace fresco {

template <int D, int DV>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out,
                                                        int Lq, int Lk, int dv_real, float scale_log2) {
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
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = exp2f(s[r] - m_new);
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

template <int D, int DV>
static int launch_attn32(const float* q, const float* k, const float* v, float* out, int B, int Lq, int Lk, int dv,
                         float scale, hipStream_t st) {
    ProfScope ps(FRESCO_PROF_ATTN_F32, B, Lq, Lk, D, st);
    hipLaunchKernelGGL((attn_f32_kernel<D, DV>), dim3((Lq + 127) / 128, B), dim3(256), 0, st, q, k, v, out, Lq, Lk,
                       dv, scale * 1.4426950408889634f);
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
