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

// ---------------------------------------------------------------------------------------------------------------
// N <= 32 frames: the per-pixel N x N attention on the matrix pipe.
//
// A block owns PB consecutive trajectories of one CFG half: R = PB*N rows each of Q, K and V.  The rows arrive by
// LDS-DMA (global_load_lds_dwordx4: one instruction per gathered row and 64 chunks, every load of the block in
// flight at once, no staging registers) into LDS rows of RS = 16 * (C/8 | 1) bytes: an odd number of 16-byte
// chunks, so that the 16 rows of an MFMA operand tile start in 16 different bank quads (ds_read_b128 of one
// chunk per row: conflict-free) and rows 4 apart sit 16 banks apart (the 2-byte reads of the V operand).
//
// One wave owns a unit = (16-row query tile, head).  For N <= 16 a tile holds 16/N whole trajectories
// (block-diagonal: scores between different trajectories are masked off), for N <= 32 a trajectory is KT = 2
// key tiles.  Per unit:
//   S^T = K Q^T      v_mfma_f32_16x16x32_f16, A = K rows, B = Q rows (16 B per lane straight from LDS);
//                    lane (i = l%16, j = l/16) gets S^T[key 4j..4j+3][query i]
//   P^T              masked softmax down the key axis: 4 values per lane x KT tiles, two cross-lane maxima;
//                    the fp32 -> fp16 packed P^T registers ARE the B operand of the next product
//   O^T = V^T P^T    v_mfma_f32_16x16x16_f16, A = V^T (2-byte LDS reads, 4 keys per lane), one extra product
//                    with A = ones gives the softmax denominator in every lane
//   O                scaled, fp16, written over the unit's own Q segment (nobody else reads it); the block
//                    then stores whole rows.
template <int D, int KT>
__global__ __launch_bounds__(512) void temporal_mfma_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v,
    const int64_t* __restrict__ fwd_map, const uint8_t* __restrict__ mask, half_t* __restrict__ out,
    int N, int HW, int H, int PB, int chunk, float scale_log2, int64_t q_ld, int64_t k_ld, int64_t v_ld, int packed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK32 = (D + 31) / 32;  // k-steps of the score product
    constexpr int NDT = (D + 15) / 16;   // 16-wide output tiles
    // gridDim.z blocks share a tile's heads: a block stages (and writes) only the columns of its Hs heads
    const int C = H * D;
    const int Hs = H / (int)gridDim.z;
    const int col0 = blockIdx.z * Hs * D;  // first channel of this block's heads
    const int CC = Hs * D / 8;             // 16-byte chunks per staged row
    const int RS = 16 * (CC | 1);
    const int c = blockIdx.y;
    const int p0 = blockIdx.x * PB;
    const int tid = threadIdx.x;
    const int NT = blockDim.x;
    const int R = PB * N;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = NT >> 6;

    char* qs = smem;
    const char* ks = qs + (size_t)R * RS;
    const char* vs = ks + (size_t)R * RS;
    int* rows = reinterpret_cast<int*>(smem + (size_t)3 * R * RS);  // [PB][N] global row index of frame g (-1: none)
    uint8_t* msk = reinterpret_cast<uint8_t*>(rows + R);            // [PB][N][N] the trajectories' mask rows

    for (int i = tid; i < R; i += NT) {
        const int pl = i / N, g = i % N;
        const int p = p0 + pl;
        int r = -1;
        if (p < HW) {
            const int row = packed ? p : (int)fwd_map[(int64_t)g * HW + p];
            // (a row table that is not a permutation is the caller's bug; never read out of bounds)
            if (row >= 0 && row < HW) r = (packed ? g * chunk + c : c * N + g) * HW + row;
        }
        rows[i] = r;
    }
    for (int i = tid; i < R * N; i += NT) {
        const int64_t gi = (int64_t)p0 * N * N + i;
        msk[i] = gi < (int64_t)HW * N * N ? mask[gi] : 0;
    }
    __syncthreads();

    // ---- stage Q, K, V rows by DMA: wave w takes rows w, w + nwaves, ...; lane = 16-byte chunk of the row ------
    const int ppr = (CC + 63) >> 6;  // DMA instructions per row
    {
        const uint32_t lds0 =
            __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
        for (int r = wave; r < R; r += nwaves) {
            int ro = __builtin_amdgcn_readfirstlane(rows[r]);
            if (ro < 0) ro = 0;  // rows without a trajectory: load something valid, never stored
            const half_t* src[3] = {q + (int64_t)ro * q_ld + col0, k + (int64_t)ro * k_ld + col0,
                                    v + (int64_t)ro * v_ld + col0};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                for (int part = 0; part < ppr; ++part) {
                    const int cc = part * 64 + lane;
                    const uint32_t m0v = lds0 + (uint32_t)((t * R + r) * RS + part * 1024);
                    if (cc < CC) {
                        const uint32_t voff = (uint32_t)cc * 16u;
                        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src[t]),
                                     "s"(m0v)
                                     : "memory");
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // ---- units (query tile, head): per-lane geometry first (nothing of it depends on the head) ----------------
    // The block's R rows are one key tile set: KT = 1: R = (16/N)*N rows of 16/N whole trajectories, one
    // query tile; KT = 2: R = N rows of one trajectory, ceil(N/16) query tiles.
    const int li = lane & 15, lj = lane >> 4;
    const int ntiles = KT == 1 ? 1 : (N + 15) / 16;
    const int qtr = KT == 1 ? li / N : 0;  // trajectory of the query slot inside the tile
    int koff[KT], voff[KT][4], kg[KT][4];
    bool kok[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        koff[kt] = min(kt * 16 + li, R - 1) * RS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rel = kt * 16 + 4 * lj + r;
            voff[kt][r] = min(rel, R - 1) * RS;
            kok[kt][r] = rel < R && (KT == 1 ? rel / N == qtr : true);
            kg[kt][r] = KT == 1 ? rel % N : min(rel, N - 1);
        }
    }
    const half4_t ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    {
        for (int tb = 0; tb < ntiles; ++tb) {
            const int qr = tb * 16 + li;
            const bool qok = qr < R;
            const int qrow = qok ? qr : R - 1;
            bool ok[KT][4];  // msk is [trajectory][f][g] = [row][g]
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ok[kt][r] = qok && kok[kt][r] && msk[qrow * N + kg[kt][r]] != 0;
            for (int h = wave; h < Hs; h += nwaves) {
                char* qseg = qs + (size_t)qrow * RS + h * D * 2;
                const char* kh = ks + h * D * 2;
                const char* vh = vs + h * D * 2;
                // every LDS read of the unit is issued before the first product
                half8_t bq[NK32], ak[KT][NK32];
#pragma unroll
                for (int s = 0; s < NK32; ++s) {
                    const int ch = s * 4 + lj;  // chunks past the head's D/8: a zero on the Q side is enough
                    const half8_t tq = *reinterpret_cast<const half8_t*>(qseg + min(ch, D / 8 - 1) * 16);
                    bq[s] = ch < D / 8 ? tq : zero8;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        ak[kt][s] = *reinterpret_cast<const half8_t*>(kh + koff[kt] + min(ch, D / 8 - 1) * 16);
                }
                half4_t av[NDT][KT];  // V^T: channel d = dt*16 + i (channels past D: any finite row, never stored)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const int d = min(dt * 16 + li, D - 1);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            av[dt][kt][r] = *reinterpret_cast<const half_t*>(vh + voff[kt][r] + d * 2);
                }
                float sc[KT][4];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < NK32; ++s)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak[kt][s], bq[s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {  // acc[r] = S^T[key 4j + r][query i]
                        sc[kt][r] = ok[kt][r] ? acc[r] * scale_log2 : -INFINITY;
                        m = fmaxf(m, sc[kt][r]);
                    }
                }
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                // (a query whose keys are all masked: exp2(-inf - -inf) = NaN, as the reference's softmax of an
                // all -inf row)
                half4_t pb[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[kt][r] = (half_t)__builtin_amdgcn_exp2f(sc[kt][r] - m);
                floatx4 den = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    den = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pb[kt], den, 0, 0, 0);
                floatx4 o[NDT];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    o[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(av[dt][kt], pb[kt], o[dt], 0, 0, 0);
                }
                const float inv = __builtin_amdgcn_rcpf(den[0]);
                // O^T[d = dt*16 + 4j + r][query i]: 4 consecutive channels of the query's row
                if (qok) {
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const int d0 = dt * 16 + 4 * lj;
                        if (d0 < D) {
                            half4_t ov;
#pragma unroll
                            for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dt][r] * inv);
                            *reinterpret_cast<half4_t*>(qseg + d0 * 2) = ov;
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- store whole output rows ---------------------------------------------------------------------
        for (int r = wave; r < R; r += nwaves) {
            const int ro = __builtin_amdgcn_readfirstlane(rows[r]);
            if (ro < 0) continue;
            half_t* dst = out + (int64_t)ro * C + col0;
            for (int part = 0; part < ppr; ++part) {
                const int cc = part * 64 + lane;
                if (cc < CC)
                    *reinterpret_cast<uint4*>(dst + cc * 8) =
                        *reinterpret_cast<const uint4*>(qs + (size_t)r * RS + cc * 16);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Longer clips (N > 32): one work item per (trajectory, query frame, head) on the vector ALUs, running-maximum
// softmax over the frames; same staging of whole rows through LDS.
template <int D>
__global__ __launch_bounds__(256, 2) void temporal_attn_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v,
    const int64_t* __restrict__ fwd_map, const uint8_t* __restrict__ mask, half_t* __restrict__ out,
    int N, int HW, int H, int PB, int chunk, float scale_log2, int64_t q_ld, int64_t k_ld, int64_t v_ld, int packed) {
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
            r = packed ? p : (int)fwd_map[(int64_t)g * HW + p];
            if (r < 0 || r >= HW) r = -1;
        }
        rows[i] = r;
    }
    // the PB*N*N mask bytes of these trajectories are contiguous in memory: staged once
    for (int i = tid; i < R * N; i += 256) {
        const int64_t gi = (int64_t)p0 * N * N + i;
        msk[i] = gi < (int64_t)HW * N * N ? mask[gi] : 0;
    }
    __syncthreads();

    const int nchunks = R * CC;
    for (int i = tid; i < nchunks; i += 256) {
        const int r = i / CC, cc = i % CC;
        const int g = r % N;
        const int row = rows[r];
        uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, vv = qv;
        if (row >= 0) {
            const int64_t b = packed ? (int64_t)g * chunk + c : (int64_t)c * N + g;
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

    for (int i = tid; i < nchunks; i += 256) {
        const int r = i / CC, cc = i % CC;
        const int g = r % N;
        const int row = rows[r];
        if (row < 0) continue;
        const int64_t b = packed ? (int64_t)g * chunk + c : (int64_t)c * N + g;
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

template <int D, int KT>
static int launch_temporal_mfma(const half_t* q, const half_t* k, const half_t* v, const int64_t* fwd_map,
                                const uint8_t* mask, half_t* out, int chunk, int N, int HW, int H, float scale,
                                int64_t q_ld, int64_t k_ld, int64_t v_ld, bool packed, hipStream_t st) {
    const int C = H * D;
    // N <= 16: one 16-row tile of 16/N trajectories per block, 4 waves share its H heads (31.5 KB of LDS at C = 320:
    // five blocks per CU, so that the row-table lookup -> DMA -> products -> stores chains of different blocks
    // overlap); N <= 32: one trajectory (two key tiles), 8 waves
    const int PB = KT == 1 ? 16 / N : 1;
    const int NT = KT == 1 ? 256 : 512;
    if ((int64_t)chunk * N * HW > 0x7fffffff) return FRESCO_EUNSUPPORTED;  // row indices are 32-bit in the kernel
    // wide rows (N <= 16): split a tile's heads over 2 or 4 blocks (each stages only its heads' columns) so that a
    // block's LDS stays near 32 KB and five of them share a CU (C = 640, N = 8: 23 -> 21 us; N = 16: 40 -> 34 us).  Not
    // for the two-key-tile form: 320-byte row pieces and one unit per wave measured 1.4x slower there.
    int hsplit = 1;
    while (KT == 1 && hsplit < 4 && H % (hsplit * 2) == 0 &&
           (size_t)PB * N * 3 * (16 * ((C / hsplit / 8) | 1)) > 40 * 1024)
        hsplit *= 2;
    const int RS = 16 * ((C / hsplit / 8) | 1);
    const size_t lds = (size_t)PB * N * (3 * RS + 4 + N);
    if (lds > 160 * 1024) return FRESCO_EUNSUPPORTED;
    if (lds > 65536)  // (the attribute is per device, and setting it is cheap: no process-global flag)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_mfma_kernel<D, KT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((HW + PB - 1) / PB, chunk, hsplit);
    ProfScope ps(FRESCO_PROF_TEMPORAL, chunk * N, HW, H, D, st);
    hipLaunchKernelGGL((temporal_mfma_kernel<D, KT>), grid, dim3(NT), lds, st, q, k, v, fwd_map, mask, out, N, HW, H,
                       PB, chunk, scale * 1.4426950408889634f, q_ld, k_ld, v_ld, packed ? 1 : 0);
    return check_launch();
}

template <int D>
static int launch_temporal(const half_t* q, const half_t* k, const half_t* v, const int64_t* fwd_map,
                           const uint8_t* mask, half_t* out, int chunk, int N, int HW, int H, float scale,
                           int64_t q_ld, int64_t k_ld, int64_t v_ld, bool packed, hipStream_t st) {
    if (N <= 16)
        return launch_temporal_mfma<D, 1>(q, k, v, fwd_map, mask, out, chunk, N, HW, H, scale, q_ld, k_ld, v_ld, packed, st);
    if (N <= 32)
        return launch_temporal_mfma<D, 2>(q, k, v, fwd_map, mask, out, chunk, N, HW, H, scale, q_ld, k_ld, v_ld, packed, st);
    const int C = H * D;
    // LDS: Q, K and V rows of PB trajectories + the row table + the mask rows; two blocks per CU where that is possible
    const size_t per_traj = (size_t)N * C * 6 + (size_t)N * 4 + (size_t)N * N;
    if (per_traj > 160 * 1024) return FRESCO_EUNSUPPORTED;
    int PB = (int)((76 * 1024) / per_traj);
    if (PB < 1) PB = 1;
    const int want = (256 + N * H - 1) / (N * H);  // enough work items for every thread
    if (PB > want) PB = want;
    const size_t lds = (size_t)PB * per_traj;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_kernel<D>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 65536 ? lds : 65536));
    dim3 grid((HW + PB - 1) / PB, chunk);
    ProfScope ps(FRESCO_PROF_TEMPORAL, chunk * N, HW, H, D, st);
    hipLaunchKernelGGL((temporal_attn_kernel<D>), grid, dim3(256), lds, st, q, k, v, fwd_map, mask, out, N, HW, H, PB,
                       chunk, scale * 1.4426950408889634f, q_ld, k_ld, v_ld, packed ? 1 : 0);
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
#define FRESCO_T_CASE(DD) \
    case DD:              \
        return launch_temporal<DD>(qh, kh, vh, fwd_map, mask, oh, chunk, N, HW, H, scale, q_ld, k_ld, v_ld, packed, st);
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
