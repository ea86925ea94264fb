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
// one write of out.  A 256-thread block owns PB trajectories of one CFG half.  Every row that crosses
// HBM does so as whole 16-byte-per-lane coalesced rows: the PB*N gathered Q, K and V rows are staged in
// LDS, work item (p, f, h) takes its 2*D-byte query segment from there into registers (packed halfs,
// v_dot2_f32_f16 against the K segments), writes its normalised result back over its own query segment,
// and the block stores whole output rows.  Work items are looped over (any N and H).
//
// Frame-parallel multi-GPU runs shard this pass by TRAJECTORY (each byte crosses the fabric once, and the
// kernel's HBM bytes shrink with the world size): fresco_temporal_pack gathers a rank's frames into
// per-destination trajectory ranges, an all-to-all delivers to every rank all N frames of its range,
// the same kernel runs on those rows without a row table ("packed" form), and the way back mirrors it
// (all-to-all, fresco_temporal_unpack).
#include "common.h"

namespace fresco {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// q, k, v rows of (batch, row) live at  base + (batch*rows_per_batch + row) * ld ; batch of (half c, frame g):
//   gather form : c * N + g                 rows_per_batch = HW, row = fwd_map[g][p]
//   packed form : g * chunk + c             rows_per_batch = P,  row = p        (q | k | v fused per row)
template <int D, bool PACKED>
__global__ __launch_bounds__(256, 2) void temporal_attn_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v,
    const int64_t* __restrict__ fwd_map, const uint8_t* __restrict__ mask, half_t* __restrict__ out,
    int N, int HW, int H, int PB, int chunk, float scale_log2, int64_t q_ld, int64_t k_ld, int64_t v_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = H * D;
    const int CC = C / 8;  // 16-byte chunks per row
    const int c = blockIdx.y;
    const int p0 = blockIdx.x * PB;
    const int tid = threadIdx.x;
    const int R = PB * N;  // staged rows per tensor

    half_t* qs = reinterpret_cast<half_t*>(smem);
    half_t* ks = qs + (size_t)R * C;
    half_t* vs = ks + (size_t)R * C;
    int* rows = reinterpret_cast<int*>(vs + (size_t)R * C);  // [PB][N] row of frame g (-1: no such trajectory)
    uint8_t* msk = reinterpret_cast<uint8_t*>(rows + R);     // [PB][N][N] the trajectories' mask rows

    for (int i = tid; i < R; i += 256) {
        const int pl = i / N, g = i % N;
        const int p = p0 + pl;
        int r = -1;
        if (p < HW) {
            r = PACKED ? p : (int)fwd_map[(int64_t)g * HW + p];
            if (r < 0 || r >= HW) r = -1;  // (a row table that is not a permutation is the caller's bug; never read out of bounds)
        }
        rows[i] = r;
    }
    // the PB*N*N mask bytes of these trajectories are contiguous in memory: staged once (a per-item read from
    // global memory inside the frame loop costs a memory round trip per frame)
    for (int i = tid; i < R * N; i += 256) {
        const int64_t gi = (int64_t)p0 * N * N + i;
        msk[i] = gi < (int64_t)HW * N * N ? mask[gi] : 0;
    }
    __syncthreads();

    // ---- stage Q, K, V rows: 16 bytes per lane, whole rows ------------------------------------------
    const int nchunks = R * CC;
    for (int i = tid; i < nchunks; i += 256) {
        const int r = i / CC, cc = i % CC;
        const int g = r % N;
        const int row = rows[r];
        uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, vv = qv;
        if (row >= 0) {
            const int64_t b = PACKED ? (int64_t)g * chunk + c : (int64_t)c * N + g;
            const int64_t ro = b * HW + row;
            qv = *reinterpret_cast<const uint4*>(q + ro * q_ld + cc * 8);
            kv = *reinterpret_cast<const uint4*>(k + ro * k_ld + cc * 8);
            vv = *reinterpret_cast<const uint4*>(v + ro * v_ld + cc * 8);
        }
        *reinterpret_cast<uint4*>(qs + (size_t)r * C + cc * 8) = qv;
        *reinterpret_cast<uint4*>(ks + (size_t)r * C + cc * 8) = kv;
        *reinterpret_cast<uint4*>(vs + (size_t)r * C + cc * 8) = vv;
    }
    __syncthreads();

    // ---- work items (trajectory pl, query frame f, head h) ------------------------------------------
    const int nwork = R * H;
    for (int w = tid; w < nwork; w += 256) {
        const int h = w % H;
        const int f = (w / H) % N;
        const int pl = w / (H * N);
        const int p = p0 + pl;
        if (p >= HW) continue;
        half_t* qseg = qs + (size_t)(pl * N + f) * C + h * D;
        half2_t qh[D / 2];
#pragma unroll
        for (int j = 0; j < D / 8; ++j) {
            const half8_t t = *reinterpret_cast<const half8_t*>(qseg + j * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) qh[j * 4 + e] = half2_t{t[2 * e], t[2 * e + 1]};
        }
        float acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = 0.f;
        float m_run = -1e30f, l_run = 0.f;
        const uint8_t* mrow = msk + (pl * N + f) * N;
        const half_t* kbase = ks + (size_t)pl * N * C + h * D;
        const half_t* vbase = vs + (size_t)pl * N * C + h * D;
        for (int g = 0; g < N; ++g) {
            if (mrow[g] == 0) continue;
            float s0 = 0.f, s1 = 0.f;  // two chains: even / odd 16-byte chunks
#pragma unroll
            for (int j = 0; j < D / 8; ++j) {
                const half8_t t = *reinterpret_cast<const half8_t*>(kbase + (size_t)g * C + j * 8);
                float& s = (j & 1) ? s1 : s0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s = __builtin_amdgcn_fdot2(qh[j * 4 + e], half2_t{t[2 * e], t[2 * e + 1]}, s, false);
            }
            const float s = (s0 + s1) * scale_log2;
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
        // the result replaces this work item's own query segment (nobody else reads it)
#pragma unroll
        for (int j = 0; j < D / 8; ++j) {
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[j * 8 + e] * inv);
            *reinterpret_cast<half8_t*>(qseg + j * 8) = o;
        }
    }
    __syncthreads();

    // ---- store whole output rows ---------------------------------------------------------------------
    for (int i = tid; i < nchunks; i += 256) {
        const int r = i / CC, cc = i % CC;
        const int g = r % N;
        const int row = rows[r];
        if (row < 0) continue;
        const int64_t b = PACKED ? (int64_t)g * chunk + c : (int64_t)c * N + g;
        *reinterpret_cast<uint4*>(out + (b * HW + row) * C + cc * 8) =
            *reinterpret_cast<const uint4*>(qs + (size_t)r * C + cc * 8);
    }
}

// Multi-GPU, way out: this rank's n_loc frames [f0, f0 + n_loc) of q, k, v, gathered along the trajectories
// into per-destination ranges:  dst[(d*n_loc + fl)*chunk + c][pl][0:3C] = (q | k | v)[c*n_loc + fl][fwd_map[f0+fl][d*Pw + pl]].
// Way back (UNPACK): out[c*n_loc + fl][fwd_map[f0+fl][d*Pw + pl]] = src[(d*n_loc + fl)*chunk + c][pl][0:C].
template <bool UNPACK>
__global__ __launch_bounds__(256) void temporal_pack_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                            const half_t* __restrict__ v,
                                                            const int64_t* __restrict__ fwd_map,
                                                            half_t* __restrict__ buf, half_t* __restrict__ out,
                                                            int chunk, int n_loc, int f0, int HW, int C, int Pw,
                                                            int64_t q_ld, int64_t k_ld, int64_t v_ld) {
    const int CC = C / 8;
    const int nt = UNPACK ? 1 : 3;
    const int fl = blockIdx.y, c = blockIdx.z;
    const int64_t per_row = (int64_t)nt * CC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)HW * per_row; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i / per_row);
        const int rem = (int)(i % per_row);
        const int t = rem / CC, cc = rem % CC;
        int row = (int)fwd_map[(int64_t)(f0 + fl) * HW + p];
        if (row < 0 || row >= HW) continue;
        const int d = p / Pw, pl = p - d * Pw;
        const int64_t prow = (((int64_t)d * n_loc + fl) * chunk + c) * Pw + pl;
        const int64_t lrow = ((int64_t)c * n_loc + fl) * HW + row;
        if (UNPACK) {
            *reinterpret_cast<uint4*>(out + lrow * C + cc * 8) = *reinterpret_cast<const uint4*>(buf + prow * C + cc * 8);
        } else {
            const half_t* src = t == 0 ? q + lrow * q_ld : (t == 1 ? k + lrow * k_ld : v + lrow * v_ld);
            *reinterpret_cast<uint4*>(buf + prow * 3 * C + t * C + cc * 8) = *reinterpret_cast<const uint4*>(src + cc * 8);
        }
    }
}

template <int D, bool PACKED>
static int launch_temporal(const half_t* q, const half_t* k, const half_t* v, const int64_t* fwd_map,
                           const uint8_t* mask, half_t* out, int chunk, int N, int HW, int H, float scale,
                           int64_t q_ld, int64_t k_ld, int64_t v_ld, hipStream_t st) {
    const int C = H * D;
    // LDS: Q, K and V rows of PB trajectories + the row table; two blocks per CU where that is possible
    const size_t per_traj = (size_t)N * C * 6 + (size_t)N * 4 + (size_t)N * N;
    if (per_traj > 160 * 1024) return FRESCO_EUNSUPPORTED;
    int PB = (int)((76 * 1024) / per_traj);
    if (PB < 1) PB = 1;
    const int want = (256 + N * H - 1) / (N * H);  // enough work items for every thread
    if (PB > want) PB = want;
    const size_t lds = (size_t)PB * per_traj;
    // (the attribute is per device, and setting it is cheap: every launch, no process-global flag)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_kernel<D, PACKED>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 65536 ? lds : 65536));
    dim3 grid((HW + PB - 1) / PB, chunk);
    ProfScope ps(FRESCO_PROF_TEMPORAL, chunk * N, HW, H, D, st);
    hipLaunchKernelGGL((temporal_attn_kernel<D, PACKED>), grid, dim3(256), lds, st, q, k, v, fwd_map, mask, out, N, HW,
                       H, PB, chunk, scale * 1.4426950408889634f, q_ld, k_ld, v_ld);
    return check_launch();
}

}  // namespace fresco

using namespace fresco;

static int temporal_dispatch(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                             const uint8_t* mask, void* out, int chunk, int N, int HW, int H, int D, float scale,
                             int64_t q_ld, int64_t k_ld, int64_t v_ld, bool packed, void* stream) {
    if (!q || !k || !v || (!packed && !fwd_map) || !mask || !out) return FRESCO_EINVAL;
    if (chunk <= 0 || N <= 0 || HW <= 0 || H <= 0 || D <= 0) return FRESCO_EINVAL;
    const int64_t Cw = (int64_t)H * D;
    if (q_ld < Cw || k_ld < Cw || v_ld < Cw || q_ld % 8 || k_ld % 8 || v_ld % 8) return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    const half_t* qh = static_cast<const half_t*>(q);
    const half_t* kh = static_cast<const half_t*>(k);
    const half_t* vh = static_cast<const half_t*>(v);
    half_t* oh = static_cast<half_t*>(out);
#define FRESCO_T_CASE(DD)                                                                                          \
    case DD:                                                                                                       \
        return packed ? launch_temporal<DD, true>(qh, kh, vh, fwd_map, mask, oh, chunk, N, HW, H, scale, q_ld, k_ld, \
                                                  v_ld, st)                                                        \
                      : launch_temporal<DD, false>(qh, kh, vh, fwd_map, mask, oh, chunk, N, HW, H, scale, q_ld,    \
                                                   k_ld, v_ld, st);
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

extern "C" int fresco_temporal_attn(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                                    const uint8_t* mask, void* out, int chunk, int N, int HW, int H,
                                    int D, float scale, void* stream) {
    const int64_t Cw = (int64_t)H * D;
    return temporal_dispatch(q, k, v, fwd_map, mask, out, chunk, N, HW, H, D, scale, Cw, Cw, Cw, false, stream);
}

extern "C" int fresco_temporal_attn_ld(const void* q, const void* k, const void* v, const int64_t* fwd_map,
                                       const uint8_t* mask, void* out, int chunk, int N, int HW, int H, int D,
                                       float scale, int64_t q_ld, int64_t k_ld, int64_t v_ld, void* stream) {
    return temporal_dispatch(q, k, v, fwd_map, mask, out, chunk, N, HW, H, D, scale, q_ld, k_ld, v_ld, false, stream);
}

extern "C" int fresco_temporal_attn_packed(const void* qkv, const uint8_t* mask, void* out, int chunk, int N, int P,
                                           int H, int D, float scale, void* stream) {
    if (!qkv) return FRESCO_EINVAL;
    const int64_t Cw = (int64_t)H * D;
    const half_t* b = static_cast<const half_t*>(qkv);
    return temporal_dispatch(b, b + Cw, b + 2 * Cw, nullptr, mask, out, chunk, N, P, H, D, scale, 3 * Cw, 3 * Cw,
                             3 * Cw, true, stream);
}

static int pack_dispatch(bool unpack, const void* q, const void* k, const void* v, const int64_t* fwd_map, void* buf,
                         void* out, int chunk, int n_loc, int f0, int HW, int C, int world, int64_t q_ld, int64_t k_ld,
                         int64_t v_ld, void* stream) {
    if (!fwd_map || !buf || chunk <= 0 || n_loc <= 0 || f0 < 0 || HW <= 0 || C <= 0 || world <= 0) return FRESCO_EINVAL;
    if (C % 8 != 0 || HW % world != 0) return FRESCO_EINVAL;
    if (unpack ? !out : (!q || !k || !v || q_ld < C || k_ld < C || v_ld < C || q_ld % 8 || k_ld % 8 || v_ld % 8))
        return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    const int per_row = (unpack ? 1 : 3) * (C / 8);
    int gx = (int)(((int64_t)HW * per_row + 255) / 256);
    if (gx > 4096) gx = 4096;
    dim3 grid(gx, n_loc, chunk);
    const half_t* qh = static_cast<const half_t*>(q);
    const half_t* kh = static_cast<const half_t*>(k);
    const half_t* vh = static_cast<const half_t*>(v);
    if (unpack)
        hipLaunchKernelGGL((temporal_pack_kernel<true>), grid, dim3(256), 0, st, qh, kh, vh, fwd_map,
                           static_cast<half_t*>(buf), static_cast<half_t*>(out), chunk, n_loc, f0, HW, C, HW / world,
                           q_ld, k_ld, v_ld);
    else
        hipLaunchKernelGGL((temporal_pack_kernel<false>), grid, dim3(256), 0, st, qh, kh, vh, fwd_map,
                           static_cast<half_t*>(buf), static_cast<half_t*>(out), chunk, n_loc, f0, HW, C, HW / world,
                           q_ld, k_ld, v_ld);
    return check_launch();
}

extern "C" int fresco_temporal_pack(const void* q, const void* k, const void* v, const int64_t* fwd_map, void* buf,
                                    int chunk, int n_loc, int f0, int HW, int C, int world, int64_t q_ld,
                                    int64_t k_ld, int64_t v_ld, void* stream) {
    return pack_dispatch(false, q, k, v, fwd_map, buf, nullptr, chunk, n_loc, f0, HW, C, world, q_ld, k_ld, v_ld,
                         stream);
}

extern "C" int fresco_temporal_unpack(const void* buf, const int64_t* fwd_map, void* out, int chunk, int n_loc, int f0,
                                      int HW, int C, int world, void* stream) {
    return pack_dispatch(true, nullptr, nullptr, nullptr, fwd_map, const_cast<void*>(buf), out, chunk, n_loc, f0, HW, C,
                         world, C, C, C, stream);
}
