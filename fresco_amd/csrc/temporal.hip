// Temporal-guided (FLATTEN) attention, fused: gather along the flow trajectory -> masked N x N
// softmax per pixel and head -> scatter back (reference: src/diffusion_hacked.py:309-367).
//
// The reference permutes K, Q, V into trajectory order with three rearrange+gather round trips,
// runs an SDPA over 2*HW*heads problems of sequence length N and gathers back with the inverse
// permutation.  Because bwd_mapping = argsort(fwd_mapping) (src/flow_utils.py:135), the whole pass is:
// for every trajectory p, frame f:  row(f) = fwd_map[f][p];  out[f][row(f)] = attention of
// q[f][row(f)] over { k[g][row(g)], v[g][row(g)] : g in frames, mask[p][f][g] }.
//
// HBM-bound (arithmetic intensity ~ N/2 flop/byte): algorithmic traffic is one read of q, k, v and
// one write of out.  A 256-thread block owns PB trajectories of one CFG half: the PB*N gathered K
// and V rows (whole rows of H*D halfs, 16-byte coalesced) are staged in LDS once and shared by the
// N query frames; thread (p, f, h) keeps its q row segment and the fp32 accumulator in registers.
#include "common.h"

namespace fresco {

template <int D>
__global__ __launch_bounds__(256) void temporal_attn_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v,
    const int64_t* __restrict__ fwd_map, const uint8_t* __restrict__ mask, half_t* __restrict__ out,
    int N, int HW, int H, int PB, float scale_log2, int n_loc, int f0, int k_rank_stride,
    int v_rank_stride, int64_t q_ld, int64_t k_ld, int64_t v_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = H * D;
    const int CC = C / 8;  // 16-byte chunks per row
    const int c = blockIdx.y;
    const int p0 = blockIdx.x * PB;
    const int tid = threadIdx.x;

    half_t* ks = reinterpret_cast<half_t*>(smem);
    half_t* vs = ks + (size_t)PB * N * C;
    int* rows = reinterpret_cast<int*>(vs + (size_t)PB * N * C);  // [PB][N] gathered row of frame g

    for (int i = tid; i < PB * N; i += 256) {
        const int pl = i / N, g = i % N;
        const int p = p0 + pl;
        rows[i] = p < HW ? (int)fwd_map[(int64_t)g * HW + p] : -1;
    }
    __syncthreads();

    const int nchunks = PB * N * CC;
    for (int i = tid; i < nchunks; i += 256) {
        const int r = i / CC, cc = i % CC;
        const int g = r % N;
        const int row = rows[r];
        uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
        if (row >= 0) {
            // frame g lives on shard g / n_loc as local frame g % n_loc (single GPU: n_loc = N)
            const int sh = g / n_loc, gl = g - sh * n_loc;
            kv = *reinterpret_cast<const uint4*>(
                k + (((int64_t)(sh * k_rank_stride + c * n_loc + gl)) * HW + row) * k_ld + cc * 8);
            vv = *reinterpret_cast<const uint4*>(
                v + (((int64_t)(sh * v_rank_stride + c * n_loc + gl)) * HW + row) * v_ld + cc * 8);
        }
        *reinterpret_cast<uint4*>(ks + (size_t)r * C + cc * 8) = kv;
        *reinterpret_cast<uint4*>(vs + (size_t)r * C + cc * 8) = vv;
    }
    __syncthreads();

    const int h = tid % H;
    const int fl = (tid / H) % n_loc;  // local query frame
    const int f = f0 + fl;             // its global frame index
    const int pl = tid / (H * n_loc);
    const int p = p0 + pl;
    if (pl >= PB || p >= HW) return;

    const int myrow = rows[pl * N + f];
    const int64_t qoff = (((int64_t)(c * n_loc + fl)) * HW + myrow) * q_ld + h * D;
    const int64_t ooff = (((int64_t)(c * n_loc + fl)) * HW + myrow) * C + h * D;
    float qf[D];
#pragma unroll
    for (int j = 0; j < D / 8; ++j) {
        const half8_t t = *reinterpret_cast<const half8_t*>(q + qoff + j * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[j * 8 + e] = (float)t[e];
    }
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const uint8_t* mrow = mask + ((int64_t)p * N + f) * N;
    const half_t* kbase = ks + (size_t)pl * N * C + h * D;
    const half_t* vbase = vs + (size_t)pl * N * C + h * D;
    for (int g = 0; g < N; ++g) {
        if (mrow[g] == 0) continue;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < D / 8; ++j) {
            const half8_t t = *reinterpret_cast<const half8_t*>(kbase + (size_t)g * C + j * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qf[j * 8 + e], (float)t[e], s);
        }
        s *= scale_log2;
        const float m_new = fmaxf(m_run, s);
        const float alpha = exp2f(m_run - m_new);
        const float pw = exp2f(s - m_new);
        m_run = m_new;
        l_run = fmaf(l_run, alpha, pw);
#pragma unroll
        for (int j = 0; j < D / 8; ++j) {
            const half8_t t = *reinterpret_cast<const half8_t*>(vbase + (size_t)g * C + j * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j * 8 + e] = fmaf(acc[j * 8 + e], alpha, pw * (float)t[e]);
        }
    }
    const float inv = 1.f / l_run;
#pragma unroll
    for (int j = 0; j < D / 8; ++j) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[j * 8 + e] * inv);
        *reinterpret_cast<half8_t*>(out + ooff + j * 8) = o;
    }
}

template <int D>
static int launch_temporal(const half_t* q, const half_t* k, const half_t* v, const int64_t* fwd_map,
                           const uint8_t* mask, half_t* out, int chunk, int N, int HW, int H,
                           float scale, int n_loc, int f0, int krs, int vrs, int64_t q_ld, int64_t k_ld,
                           int64_t v_ld, hipStream_t st) {
    const int tpp = n_loc * H;  // threads per trajectory
    if (tpp > 256) return FRESCO_EUNSUPPORTED;
    int PB = 256 / tpp;
    const int C = H * D;
    // LDS: K and V rows (2 * PB*N*C halfs) + row table; keep two blocks per CU resident
    while (PB > 1 && (size_t)PB * N * C * 4 + PB * N * 4 > 72 * 1024) PB >>= 1;
    const size_t lds = (size_t)PB * N * C * 4 + (size_t)PB * N * 4;
    if (lds > 160 * 1024) return FRESCO_EUNSUPPORTED;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_kernel<D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    dim3 grid((HW + PB - 1) / PB, chunk);
    ProfScope ps(FRESCO_PROF_TEMPORAL, chunk * N, HW, H, D, st);
    hipLaunchKernelGGL((temporal_attn_kernel<D>), grid, dim3(256), lds, st, q, k, v, fwd_map, mask, out,
                       N, HW, H, PB, scale * 1.4426950408889634f, n_loc, f0, krs, vrs, q_ld, k_ld, v_ld);
    return check_launch();
}

}  // namespace fresco

using namespace fresco;

static int temporal_dispatch(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                             const uint8_t* mask, void* out, int chunk, int N, int HW, int H, int D, float scale,
                             int n_loc, int f0, int k_rank_stride, int v_rank_stride, int64_t q_ld, int64_t k_ld,
                             int64_t v_ld, void* stream) {
    if (!q || !k || !v || !fwd_map || !mask || !out) return FRESCO_EINVAL;
    if (chunk <= 0 || N <= 0 || HW <= 0 || H <= 0 || D <= 0) return FRESCO_EINVAL;
    if (n_loc <= 0 || N % n_loc != 0 || f0 < 0 || f0 + n_loc > N || f0 % n_loc != 0) return FRESCO_EINVAL;
    if (k_rank_stride < 0 || v_rank_stride < 0) return FRESCO_EINVAL;
    const int64_t Cw = (int64_t)H * D;
    if (q_ld < Cw || k_ld < Cw || v_ld < Cw || q_ld % 8 || k_ld % 8 || v_ld % 8) return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    const half_t* qh = static_cast<const half_t*>(q);
    const half_t* kh = static_cast<const half_t*>(k);
    const half_t* vh = static_cast<const half_t*>(v);
    half_t* oh = static_cast<half_t*>(out);
#define FRESCO_T_CASE(DD) \
    case DD:              \
        return launch_temporal<DD>(qh, kh, vh, fwd_map, mask, oh, chunk, N, HW, H, scale, n_loc, f0, \
                                   k_rank_stride, v_rank_stride, q_ld, k_ld, v_ld, st);
    switch (D) {
        FRESCO_T_CASE(8)
        FRESCO_T_CASE(16)
        FRESCO_T_CASE(32)
        FRESCO_T_CASE(40)
        FRESCO_T_CASE(64)
        FRESCO_T_CASE(80)
        default:
            return FRESCO_EUNSUPPORTED;
    }
#undef FRESCO_T_CASE
}

extern "C" int fresco_temporal_attn_sharded(const void* q, const void* k, const void* v,
                                            const int64_t* fwd_map, const uint8_t* mask, void* out,
                                            int chunk, int N, int HW, int H, int D, float scale, int n_loc,
                                            int f0, int k_rank_stride, int v_rank_stride, void* stream) {
    const int64_t Cw = (int64_t)H * D;
    return temporal_dispatch(q, k, v, fwd_map, mask, out, chunk, N, HW, H, D, scale, n_loc, f0, k_rank_stride,
                             v_rank_stride, Cw, Cw, Cw, stream);
}

extern "C" int fresco_temporal_attn(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                                    const uint8_t* mask, void* out, int chunk, int N, int HW, int H,
                                    int D, float scale, void* stream) {
    const int64_t Cw = (int64_t)H * D;
    return temporal_dispatch(q, k, v, fwd_map, mask, out, chunk, N, HW, H, D, scale, N, 0, 0, 0, Cw, Cw, Cw,
                             stream);
}

extern "C" int fresco_temporal_attn_ld(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                                       const uint8_t* mask, void* out, int chunk, int N, int HW, int H, int D,
                                       float scale, int64_t q_ld, int64_t k_ld, int64_t v_ld, void* stream) {
    return temporal_dispatch(q, k, v, fwd_map, mask, out, chunk, N, HW, H, D, scale, N, 0, 0, 0, q_ld, k_ld,
                             v_ld, stream);
}
