// Dense fp16 attention with shared key groups for FRESCO's spatial-guided and efficient
// cross-frame passes (reference: src/diffusion_hacked.py:225-247, 250-254, 281-285, 303-305, 371).
//
// Two kernels:
//   kv_pack_kernel  : gathers the selected K / V rows of one key group and writes them in the
//                     tile order the MFMA loop consumes:  Kp[g][h][Mpad][DPK]  (rows = keys, head
//                     dim zero-padded to a multiple of 16) and  Vt[g][h][DPV][Mpad]  (V transposed:
//                     rows = head dim padded to a multiple of 32, keys contiguous).  HBM-bound.
//   attn_flash_kernel: flash-style attention on v_mfma_f32_32x32x16_f16.  One wave owns 32 query
//                     rows; S^T = K Q^T is computed "swapped" so that every lane holds the scores
//                     of ONE query (its column of the 32x32 MFMA C tile): the exponentiated scores
//                     are already laid out as the B operand of  O^T = V^T P^T.
//                     K / V^T tiles of 64 keys are DMA'd into double-buffered LDS whose row strides
//                     are odd multiples of 16 B (conflict-free ds_read_b128).
//                     The softmax bookkeeping rides in the MFMAs wherever the head dim leaves room:
//                     a ones ROW in V^T makes the PV product deliver the row sum, a ones COLUMN in K
//                     against -m in Q's spare column makes the QK product subtract the running max.
//                     Per wave, from the key norms kv_pack records (Cauchy-Schwarz bound on the logits):
//                     the max search is dropped when no exponent can leave fp16 range, and the exponent
//                     scale is folded into the fp16 Q only while that costs no more than P's own rounding.
//
// MFMA 32x32x16 f16 operand layout used below (gfx950): lane l supplies 8 consecutive k for
// row/col (l & 31), k-chunk (l >> 5); C/D: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5).
#include "common.h"
#include <type_traits>

namespace fresco {

template <int D>
struct AttnCfg {
    static constexpr int DPK = (D + 15) / 16 * 16;  // head dim padded for the QK^T contraction
    static constexpr int DPV = (D + 31) / 32 * 32;  // head dim padded for the O^T row blocks
    static constexpr int NKS = DPK / 16;            // MFMA k-steps per QK^T block
    static constexpr int NDB = DPV / 32;            // 32-row blocks of O^T
    static constexpr int NKC = DPK / 8;             // 16-byte chunks per K row
    static constexpr int KROW = DPK * 2 + (((DPK * 2 / 16) % 2 == 0) ? 16 : 0);  // LDS bytes per K row
    static constexpr int VROW = 64 * 2 + 16;                                     // LDS bytes per V^T row
    static constexpr int KCR = KROW / 16;  // 16-byte chunks per LDS K row (data + pad)
    static constexpr int VCR = VROW / 16;
    static constexpr int KTILE = 64 * KROW;                          // multiple of 1 KiB (64 rows)
    static constexpr int VTILE = (DPV * VROW + 1023) / 1024 * 1024;  // rounded up: whole 1 KiB DMA pieces
    static constexpr int KDMA = KTILE / 1024;                        // wave-level DMA instructions per tile
    static constexpr int VDMA = VTILE / 1024;
    static constexpr int NP = KDMA + VDMA;      // 1 KiB DMA pieces per tile
    static constexpr int PW = (NP + 3) / 4;     // pieces per wave and tile (the last ones may be pad pieces)
    static constexpr int BUFB = PW * 4 * 1024;  // LDS bytes per buffer: K tile, V^T tile, pad pieces
    static constexpr int LDS_BYTES = 2 * BUFB;
    static constexpr int KCH = 64 * NKC;  // 16-byte chunks in a K tile
    static constexpr int VCH = DPV * 8;   // 16-byte chunks in a V^T tile
    static constexpr bool ONES = DPV > D;  // spare V^T row D holds ones: the PV MFMA also yields the row sum
    static constexpr bool MCOL = DPK > D;  // spare K column D holds ones: Q column D carries -m_run, so the
                                           // QK MFMA subtracts the running max (no C operand to keep around)
    static constexpr int KPT = (KCH + 255) / 256;
    static constexpr int VPT = (VCH + 255) / 256;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// Ablation switch for tools/ablate_attn.hip (timing experiments only; the product build uses 0):
// 1 = no exp (P = exponent argument), 2 = no softmax VALU at all, 3 = no PV MFMAs, 4 = no QK MFMAs,
// 5 = no K/V staging (tile 0 reused, no barrier), 6 = no LDS fragment reads (constant fragments)
#ifndef FRESCO_ABL
#define FRESCO_ABL 0
#endif

// online softmax: skip the O rescale while the tile max grows by less than this (log2 units);
// P then reaches at most 2^8 = 256, far inside fp16 range, and stays exactly normalised by the row sum
#define RESCALE_THR 8.0f
// no running max at all when |c q| max|k| - m_run stays below this (P <= 2^14 = 16384 < 65504)
#define NOMAX_THR 14.0f
// largest |exponent| (log2 units) for which the scale is folded into the fp16 Q
#define FOLD_MAX 16.0f

static inline int mpad_of(int M) { return (M + 63) / 64 * 64; }

// ---------------------------------------------------------------------------------------------
// pack: grid (Mpad/64, H, G), 256 threads
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void kv_pack_kernel(const half_t* __restrict__ k,
                                                       const half_t* __restrict__ v,
                                                       const int32_t* __restrict__ kv_rows,
                                                       half_t* __restrict__ kp, half_t* __restrict__ vt,
                                                       float* __restrict__ ktmax, int H, int M, int Mpad,
                                                       int64_t group_rows, int64_t kv_ld) {
    using Cfg = AttnCfg<D>;
    const int tile = blockIdx.x, h = blockIdx.y, g = blockIdx.z;
    __shared__ int32_t rows[64];
    __shared__ __attribute__((aligned(16))) half_t vs[64][D + 8];  // +8 halfs: 16-B aligned rows

    if (threadIdx.x < 64) {
        const int m = tile * 64 + threadIdx.x;
        int32_t r = -1;
        if (m < M) r = kv_rows ? kv_rows[m] : m;
        rows[threadIdx.x] = r;
    }
    __syncthreads();

    const uint4 zero = make_uint4(0, 0, 0, 0);
    // K: 64 keys x NKC chunks, written contiguously
    half_t* kdst = kp + ((int64_t)(g * H + h) * Mpad + tile * 64) * Cfg::DPK;
    for (int c = threadIdx.x; c < 64 * Cfg::NKC; c += 256) {
        const int row = c / Cfg::NKC, dc = c % Cfg::NKC;
        uint4 val = zero;
        const int32_t r = rows[row];
        if (r >= 0 && dc * 8 < D)
            val = *reinterpret_cast<const uint4*>(k + ((int64_t)g * group_rows + r) * kv_ld + h * D + dc * 8);
        else if (Cfg::MCOL && r >= 0 && dc * 8 == D)
            val.x = 0x3C00u;  // K[key][D] = 1.0: with Q[query][D] = -m the QK MFMA delivers  q.k - m
        *reinterpret_cast<uint4*>(kdst + (int64_t)c * 8) = val;
    }
    // largest squared key norm of the tile (one thread per key, fixed summation order): the flash kernel
    // bounds every logit of a query by |q| max|k| (Cauchy-Schwarz) and drops the running-max search when
    // that bound cannot leave fp16 range
    {
        // four threads per key, each a fixed subset of the 16-byte chunks; combined in a fixed order
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
        const int32_t r = rows[row];
        float n2 = 0.f;
        if (r >= 0) {
            const half_t* kr = k + ((int64_t)g * group_rows + r) * kv_ld + h * D;
            for (int dc = part; dc < D / 8; dc += 4) {
                const half8_t kk = *reinterpret_cast<const half8_t*>(kr + dc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) n2 = fmaf((float)kk[e], (float)kk[e], n2);
            }
        }
        n2 += __shfl_xor(n2, 1, 64);
        n2 += __shfl_xor(n2, 2, 64);
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) n2 = fmaxf(n2, __shfl_xor(n2, off, 64));
        __shared__ float wmax[4];
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = n2;
        __syncthreads();
        if (threadIdx.x == 0)
            ktmax[(int64_t)(g * H + h) * (Mpad / 64) + tile] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    }
    // V: stage the 64 x D slab, then write it transposed
    for (int c = threadIdx.x; c < 64 * (D / 8); c += 256) {
        const int row = c / (D / 8), dc = c % (D / 8);
        uint4 val = zero;
        const int32_t r = rows[row];
        if (r >= 0)
            val = *reinterpret_cast<const uint4*>(v + ((int64_t)g * group_rows + r) * kv_ld + h * D + dc * 8);
        *reinterpret_cast<uint4*>(&vs[row][dc * 8]) = val;
    }
    __syncthreads();
    half_t* vdst = vt + (int64_t)(g * H + h) * Cfg::DPV * Mpad + tile * 64;
    for (int c = threadIdx.x; c < Cfg::DPV * 8; c += 256) {
        const int d = c >> 3, kc = c & 7;
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t val = (half_t)0;
            if (d < D)
                val = vs[kc * 8 + j][d];
            else if (Cfg::ONES && d == D && rows[kc * 8 + j] >= 0)
                val = (half_t)1;  // ones row: only real keys count towards the softmax denominator
            o[j] = val;
        }
        *reinterpret_cast<half8_t*>(vdst + (int64_t)d * Mpad + kc * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// flash attention: grid (H * nQblk * B), 256 threads = 4 waves x QB blocks of 32 query rows
// blockIdx.x = (b * nQblk + qblk) * H + h   -> head h lands on XCD (h % 8): each XCD's L2 holds
// only its own heads' packed K / V^T.
// QB query blocks per wave (the product instantiates QB = 1, see launch_attn).
// ---------------------------------------------------------------------------------------------
template <int D, int QB, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_flash_kernel(const half_t* __restrict__ q,
                                                          const half_t* __restrict__ kp,
                                                          const half_t* __restrict__ vt,
                                                          const float* __restrict__ ktmax,
                                                          half_t* __restrict__ out, int B, int H, int Lq,
                                                          int M, int Mpad, int batch_per_group,
                                                          float scale_log2, float diag_bias_log2, int64_t q_ld) {
    using Cfg = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 128 * QB;  // query rows per workgroup

    const int nQblk = (Lq + ROWS - 1) / ROWS;
    const int h = blockIdx.x % H;
    const int qblk = (blockIdx.x / H) % nQblk;
    const int b = blockIdx.x / (H * nQblk);
    const int g = b / batch_per_group;
    const int C = H * D;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow0 = qblk * ROWS + wave * 32 * QB + l31;  // row of query block 0; block j: + 32*j
    // S^T row (lane & 31) is fed with key  swap_bits_2_3(lane & 31): the C-tile registers of a
    // lane then hold keys 16*(r>>3) + 8*hi + (r&7), i.e. 8 consecutive keys per MFMA k-chunk.
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

    // Q fragments (B operand of S^T = K Q^T), resident for the whole kernel
    half8_t qf[QB][Cfg::NKS];
    float q2[QB];  // |q|^2 of this lane's query
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int qr = qrow0 + 32 * j;
        const half_t* qp = q + ((int64_t)b * Lq + (qr < Lq ? qr : Lq - 1)) * q_ld + h * D;
        q2[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < Cfg::NKS; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            half8_t t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (d0 < D) t = *reinterpret_cast<const half8_t*>(qp + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) q2[j] = fmaf((float)t[e], (float)t[e], q2[j]);
            qf[j][ks] = t;
        }
        q2[j] += __shfl_xor(q2[j], 32, 64);
    }

    // Cauchy-Schwarz: every logit of this lane's query is bounded by |q| max|k| (max|k|^2 per key tile comes
    // from kv_pack).  Two per-wave decisions hang on it:
    //  * FOLDED scale: the exponent scale c = softmax scale * log2 e is multiplied into Q once (one fp16
    //    rounding of c*q) and the MFMA delivers exponent arguments directly.  That rounding perturbs an
    //    exponent by at most 2^-12 * c|q||k|, so it is taken only while c|q||k| <= FOLD_MAX (error of the
    //    order of P's own fp16 rounding); otherwise Q stays exact and every score is multiplied by c in fp32.
    //  * no running-max search (NOMAX, below) when the bound cannot leave fp16 range.
    // Accumulator units u: exponent argument = cmul * u, with (qs, cmul) = (c, 1) folded or (1, c) exact.
    float kmax;
    {
        const int nTk = Mpad / 64;
        const float* km = ktmax + (int64_t)(g * H + h) * nTk;
        float k2 = 0.f;
        for (int i = lane; i < nTk; i += 64) k2 = fmaxf(k2, km[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) k2 = fmaxf(k2, __shfl_xor(k2, off, 64));
        kmax = sqrtf(k2);
    }
    bool fold_ok = true;
#pragma unroll
    for (int j = 0; j < QB; ++j) fold_ok = fold_ok && (scale_log2 * sqrtf(q2[j]) * kmax <= FOLD_MAX);
    // (readfirstlane: tells the compiler the vote is wave-uniform, so the paths below are scalar branches)
    const bool folded = __builtin_amdgcn_readfirstlane((int)__all(fold_ok)) != 0;
    const float qs = folded ? scale_log2 : 1.f;
    const float cmul = folded ? 1.f : scale_log2;
    float qbound[QB];  // bound on the accumulators (units u), with a margin for the roundings above
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        qbound[j] = qs * sqrtf(q2[j]) * kmax * 1.001f + 1e-3f;
        if (folded) {
#pragma unroll
            for (int ks = 0; ks < Cfg::NKS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[j][ks][e] = (half_t)((float)qf[j][ks][e] * scale_log2);
        }
    }
    const float resc_thr = RESCALE_THR / cmul;  // thresholds and the diagonal bias in accumulator units
    const float diag_u = diag_bias_log2 / cmul;

    const char* kg = reinterpret_cast<const char*>(kp + (int64_t)(g * H + h) * Mpad * Cfg::DPK);
    const char* vg = reinterpret_cast<const char*>(vt + (int64_t)(g * H + h) * Cfg::DPV * Mpad);
    const int nT = Mpad / 64;

    // ---- staging: global -> LDS by DMA (global_load_lds_dwordx4), no register round trip ---------
    // One wave-level instruction fills 1 KiB of LDS: destination = wave-uniform base + lane * 16, the
    // source address is per lane.  The K tile and the V^T tile are contiguous in LDS and both whole
    // KiB, so a tile is NP pieces (rounded up to 4 per round); piece p = i*4 + wave is issued by wave p % 4.  Everything that does
    // not change from tile to tile is computed once: a per-lane byte offset inside the packed image
    // (the pad chunks of the LDS row layout point at a valid dummy source) and a wave-uniform running
    // base that advances by one tile per issue -- the loop body carries scalar adds and the DMA
    // instructions only (no per-tile address VALU, no exec-mask branches).
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    uint64_t dma_addr[Cfg::PW];  // this lane's source address of piece i*4 + wave in the next tile to stage
    int dma_step[Cfg::PW];       // bytes per tile (wave-uniform): K rows are tile-contiguous, V^T advances 64 keys
#pragma unroll
    for (int i = 0; i < Cfg::PW; ++i) {
        const int p = i * 4 + wave_s;
        const bool isk = p < Cfg::KDMA;
        const int ck = p * 64 + lane;  // linear 16-byte chunk of the LDS K tile
        const int krw = ck / Cfg::KCR, kdc = ck % Cfg::KCR;
        const uint32_t koff = kdc < Cfg::NKC ? (uint32_t)(krw * Cfg::NKC + kdc) * 16u : 0u;
        const int cv = (p - Cfg::KDMA) * 64 + lane;
        const int vd = cv / Cfg::VCR, vkc = cv % Cfg::VCR;
        // pieces past the tile (p >= NP, when NP is not a multiple of 4) re-read chunk 0 into the pad KiBs
        const uint32_t voff = (vd < Cfg::DPV && vkc < 8) ? (uint32_t)(vd * Mpad + vkc * 8) * 2u : 0u;
        dma_addr[i] = reinterpret_cast<uint64_t>(isk ? kg : vg) + (isk ? koff : voff);
        dma_step[i] = isk ? Cfg::KCH * 16 : 128;
    }
    auto stage_next = [&](int buf) __attribute__((always_inline)) {  // issues the next not-yet-staged tile into LDS buffer `buf`
        char* dst = smem + buf * Cfg::BUFB + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < Cfg::PW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_addr[i]),
                                             (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, 0, 0);
            dma_addr[i] += dma_step[i];
        }
    };

    floatx16 o[QB][Cfg::NDB];
    // The accumulators must come out as  c*s - m_run  (no per-score subtraction).  MCOL: -m_run rides in Q's
    // spare column D against the ones column of the packed K (m_run is kept on the fp16 grid so that the
    // value the MFMA subtracts is exactly the one the rescale factors are computed from).  Otherwise
    // -m_run sits in all 16 registers of `negm`, the C operand of the first QK MFMA.
    constexpr int MKS = Cfg::MCOL ? D / 16 : 0, MHI = (D % 16) / 8, ME = D % 8;
    floatx16 negm[QB];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m_run[j] = 0.f;  // reference point of the exponent (log2 domain); tile 0 moves it to the row max
        l_run[j] = 0.f;  // row sum when V^T has no spare row for the ones-trick (this lane's keys)
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[j][r] = 0.f;
#pragma unroll
        for (int db = 0; db < Cfg::NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][db][r] = 0.f;
    }

    stage_next(0);
    __syncthreads();

    const bool need_diag = diag_bias_log2 != 0.f;

    // One 64-key tile.  FIX = true adds the per-element fix-ups (padded keys of the last tile,
    // diagonal bias); it is a separate instantiation so that the common path carries none of it.
    // NOMAX = true (only after tile 0 has anchored m_run, and only when `qbound` proves that no exponent
    // argument can exceed NOMAX_THR): the running-max search and the rescale test are dropped -- P is then
    // at most 2^NOMAX_THR, inside fp16 range, and the row sum normalises it exactly as before.
    // EXACT = the wave keeps Q unscaled: scores are multiplied by c in fp32 before the exponential.
    auto tile = [&](int t, auto fix_c, auto nomax_c, auto exact_c) __attribute__((always_inline)) {
        constexpr bool FIX = decltype(fix_c)::value;
        constexpr bool NOMAX = decltype(nomax_c)::value;
        constexpr bool EXACT = decltype(exact_c)::value;
        const float cm = EXACT ? cmul : 1.f;
        const int buf = (FRESCO_ABL == 5) ? 0 : (t & 1);
        const char* kb = smem + buf * Cfg::BUFB;
        const char* vb = kb + Cfg::KTILE;

        // ---- S^T = K Q^T : per query block two independent 32-key accumulators ------------------
        floatx16 s[QB][2];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[j][0][r] = Cfg::MCOL ? 0.f : negm[j][r];
                s[j][1][r] = Cfg::MCOL ? 0.f : negm[j][r];
            }
        }
        const char* kr = kb + krow * Cfg::KROW + hi * 16;
#pragma unroll
        for (int ks = 0; ks < Cfg::NKS; ++ks) {
            half8_t a0, a1;
            if (FRESCO_ABL == 6) {
                a0 = qf[0][ks];
                a1 = qf[0][ks];
            } else {
                a0 = *reinterpret_cast<const half8_t*>(kr + ks * 32);
                a1 = *reinterpret_cast<const half8_t*>(kr + 32 * Cfg::KROW + ks * 32);
            }
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                if (FRESCO_ABL == 4) {
                    s[j][0][ks] += (float)a0[0];
                    s[j][1][ks] += (float)a1[0];
                } else {
                    s[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[j][ks], s[j][0], 0, 0, 0);
                    s[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[j][ks], s[j][1], 0, 0, 0);
                }
            }
        }

        // every wave has left tile t-1 (barrier below), so its buffer can be refilled while tile t runs;
        // issued here, the scalar adds + DMA instructions sit in the shadow of the QK MFMAs
        // (the sites without fix-ups only run tiles t < nT - 1: there is always a next tile)
        if (FRESCO_ABL != 5 && (!FIX || t + 1 < nT)) stage_next(buf ^ 1);

        half8_t pf[QB][4];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            if (FRESCO_ABL == 2) {  // keep S live, skip all softmax arithmetic
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    asm volatile("" ::"v"(s[j][0][r]), "v"(s[j][1][r]));
                    pf[j][r >> 3][r & 7] = (half_t)0.01f;
                    pf[j][2 + (r >> 3)][r & 7] = (half_t)0.01f;
                }
                continue;
            }
            if (FIX) {
                const int qr = qrow0 + 32 * j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key0 = t * 64 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (need_diag && key0 == qr) s[j][0][r] += diag_u;
                    if (need_diag && key0 + 32 == qr) s[j][1][r] += diag_u;
                    if (key0 >= M) s[j][0][r] = -1e30f;
                    if (key0 + 32 >= M) s[j][1][r] = -1e30f;
                }
            }
            // ---- online softmax, one query per lane.  s = exponent argument relative to m_run; the
            // reference point moves (and O, l are rescaled) only when the tile max exceeds it by more than
            // RESCALE_THR -- or on tile 0, which anchors it at the row's first-tile max.
            float mt = 0.f;
            if (!NOMAX) {
                mt = fmaxf(s[j][0][0], s[j][1][0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, s[j][0][r]), s[j][1][r]);  // v_max3_f32
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            }
            if (!NOMAX && (t == 0 || __builtin_amdgcn_readfirstlane((int)__any(mt > resc_thr)) != 0)) {
                float delta = (t == 0) ? mt : fmaxf(mt, 0.f);
                if (Cfg::MCOL) {
                    // stays fp16-representable (and finite: logits beyond +-6e4 log2 units saturate)
                    const float m_new = (float)(half_t)fminf(fmaxf(m_run[j] + delta, -6.0e4f), 6.0e4f);
                    delta = m_new - m_run[j];
                    m_run[j] = m_new;
                    const half_t nm = (half_t)(-m_new);
                    qf[j][MKS][ME] = (hi == MHI) ? nm : qf[j][MKS][ME];
                } else {
                    m_run[j] += delta;
                }
                const float alpha = __builtin_amdgcn_exp2f(-delta * cm);
                l_run[j] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!Cfg::MCOL) negm[j][r] = -m_run[j];
                    s[j][0][r] -= delta;
                    s[j][1][r] -= delta;
                }
#pragma unroll
                for (int db = 0; db < Cfg::NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[j][db][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float x0 = EXACT ? s[j][kbk][r] * cm : s[j][kbk][r];
                    const float x1 = EXACT ? s[j][kbk][r + 1] * cm : s[j][kbk][r + 1];
                    const float p0 = (FRESCO_ABL == 1) ? x0 : __builtin_amdgcn_exp2f(x0);
                    const float p1 = (FRESCO_ABL == 1) ? x1 : __builtin_amdgcn_exp2f(x1);
                    if (!Cfg::ONES) psum += p0 + p1;
                    pf[j][kbk * 2 + (r >> 3)][r & 7] = (half_t)p0;
                    pf[j][kbk * 2 + (r >> 3)][(r & 7) + 1] = (half_t)p1;
                }
            if (!Cfg::ONES) l_run[j] += psum;
        }

        // ---- O^T += V^T P^T  (row D of V^T is all ones when it is spare: O^T[D] = row sum) ---------
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const char* vr = vb + l31 * Cfg::VROW + (kc * 16 + hi * 8) * 2;
#pragma unroll
            for (int db = 0; db < Cfg::NDB; ++db) {
                const half8_t a = (FRESCO_ABL == 6) ? qf[0][0] : *reinterpret_cast<const half8_t*>(vr + db * 32 * Cfg::VROW);
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    if (FRESCO_ABL == 3)
                        o[j][db][kc] += (float)a[0] * (float)pf[j][kc][0];
                    else
                        o[j][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[j][kc], o[j][db], 0, 0, 0);
                }
            }
        }
        if (FRESCO_ABL != 5) __syncthreads();  // also drains this wave's DMA pieces (vmcnt) before release
    };

    const std::integral_constant<bool, true> yes;
    const std::integral_constant<bool, false> no;
    // Three call sites per precision variant: search + deferred rescale from tile 0 on; without the search
    // once `qbound` allows it; the fix-up form for the last tile (padded keys) or, with a diagonal bias, for
    // every tile.
    auto run = [&](auto exact_c) __attribute__((always_inline)) {
        int t = 0;
        if (!need_diag) {
            bool nomax = false;
            for (; t < nT - 1 && !nomax; ++t) {
                tile(t, no, no, exact_c);
                if (t == 0) {
                    bool safe = true;
#pragma unroll
                    for (int j = 0; j < QB; ++j) safe = safe && (cmul * (qbound[j] - m_run[j]) <= NOMAX_THR);
                    nomax = __builtin_amdgcn_readfirstlane((int)__all(safe)) != 0;
                }
            }
            for (; t < nT - 1; ++t) tile(t, no, yes, exact_c);
        }
        for (; t < nT; ++t) tile(t, yes, no, exact_c);
    };
    if (folded)
        run(no);
    else
        run(yes);

    // ---- epilogue: normalise, store O[q][h*D + d] -------------------------------------------------
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        float l_tot;
        if (Cfg::ONES) {
            // O^T row D: C-tile row rr = D % 32 lives in register (rr&3) + 4*(rr>>3) of lanes with hi = (rr>>2)&1
            constexpr int rr = D % 32;
            l_tot = __shfl(o[j][D / 32][(rr & 3) + 4 * (rr >> 3)], l31 + 32 * ((rr >> 2) & 1), 64);
        } else {
            l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        }
        const float inv = 1.f / l_tot;
        const int qr = qrow0 + 32 * j;
        if (qr < Lq) {
            half_t* op = out + ((int64_t)b * Lq + qr) * C + h * D;
#pragma unroll
            for (int db = 0; db < Cfg::NDB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = db * 32 + g4 * 8 + hi * 4;
                    if (d0 < D) {
                        half4_t w;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) w[jj] = (half_t)(o[j][db][g4 * 4 + jj] * inv);
                        *reinterpret_cast<half4_t*>(op + d0) = w;
                    }
                }
        }
    }
}

template <int D, int QB, int MINW>
static void launch_flash(const half_t* q, const half_t* kp, const half_t* vt, half_t* out, int B, int H,
                         int Lq, int M, int Mpad, int n_groups, float scale, float diag_bias, int64_t q_ld,
                         const float* ktmax, hipStream_t st) {
    using Cfg = AttnCfg<D>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_flash_kernel<D, QB, MINW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        attr_set = true;
    }
    const int nQblk = (Lq + 128 * QB - 1) / (128 * QB);
    const float log2e = 1.4426950408889634f;
    ProfScope ps(FRESCO_PROF_ATTN_FLASH, B * H, Lq, M, D, st);
    hipLaunchKernelGGL((attn_flash_kernel<D, QB, MINW>), dim3(H * nQblk * B), dim3(256), Cfg::LDS_BYTES, st, q,
                       kp, vt, ktmax, out, B, H, Lq, M, Mpad, B / n_groups, scale * log2e, diag_bias * log2e, q_ld);
}

template <int D>
static int launch_attn(const half_t* q, const half_t* k, const half_t* v, const int32_t* kv_rows,
                       half_t* out, char* ws, int B, int H, int Lq, int n_groups, int M,
                       int64_t group_rows, float scale, float diag_bias, int64_t q_ld, int64_t kv_ld,
                       hipStream_t st) {
    using Cfg = AttnCfg<D>;
    const int Mpad = mpad_of(M);
    half_t* kp = reinterpret_cast<half_t*>(ws);
    half_t* vt = kp + (size_t)n_groups * H * Mpad * Cfg::DPK;
    float* ktmax = reinterpret_cast<float*>(vt + (size_t)n_groups * H * Mpad * Cfg::DPV);
    dim3 pg(Mpad / 64, H, n_groups);
    {
        ProfScope ps(FRESCO_PROF_KV_PACK, n_groups, H, M, D, st);
        hipLaunchKernelGGL((kv_pack_kernel<D>), pg, dim3(256), 0, st, k, v, kv_rows, kp, vt, ktmax, H, M, Mpad,
                           group_rows, kv_ld);
    }
    // One query block per wave; waves per SIMD the register allocator is asked to make room for.  Measured on
    // MI355X (round 1): two query blocks per wave (QB = 2, halves the LDS reads per MFMA) 15-25 % slower at the
    // one or two waves per SIMD they leave; one more wave per SIMD forced by launch bounds (spills) 5 % slower.
    constexpr int MINW = D <= 40 ? 3 : (D <= 80 ? 2 : 1);
    launch_flash<D, 1, MINW>(q, kp, vt, out, B, H, Lq, M, Mpad, n_groups, scale, diag_bias, q_ld, ktmax, st);
    return check_launch();
}

static size_t attn_ws_bytes(int n_groups, int H, int M, int D) {
    const size_t Mpad = mpad_of(M);
    const size_t dpk = (D + 15) / 16 * 16, dpv = (D + 31) / 32 * 32;
    return align_up((size_t)n_groups * H * Mpad * (dpk + dpv) * sizeof(half_t) +
                        (size_t)n_groups * H * (Mpad / 64) * sizeof(float), 256);
}

}  // namespace fresco

using namespace fresco;

extern "C" size_t fresco_attn_workspace_bytes(int n_groups, int H, int M, int D) {
    if (n_groups <= 0 || H <= 0 || M <= 0 || D <= 0) return 0;
    return attn_ws_bytes(n_groups, H, M, D);
}

extern "C" int fresco_attn_fwd_ld(const void* q, const void* k, const void* v, const int32_t* kv_rows,
                                  void* out, void* workspace, size_t workspace_bytes, int B, int H,
                                  int Lq, int D, int n_groups, int M, int64_t group_rows, float scale,
                                  float diag_bias, int64_t q_ld, int64_t kv_ld, void* stream) {
    if (!q || !k || !v || !out || !workspace) return FRESCO_EINVAL;
    if (q_ld < (int64_t)H * D || kv_ld < (int64_t)H * D || q_ld % 8 != 0 || kv_ld % 8 != 0) return FRESCO_EINVAL;
    if (B <= 0 || H <= 0 || Lq <= 0 || D <= 0 || n_groups <= 0 || M <= 0 || group_rows <= 0)
        return FRESCO_EINVAL;
    if (B % n_groups != 0 || !(scale > 0.f)) return FRESCO_EINVAL;
    if (workspace_bytes < attn_ws_bytes(n_groups, H, M, D)) return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const half_t* qh = static_cast<const half_t*>(q);
    const half_t* kh = static_cast<const half_t*>(k);
    const half_t* vh = static_cast<const half_t*>(v);
    half_t* oh = static_cast<half_t*>(out);
    char* ws = static_cast<char*>(workspace);
#define FRESCO_ATTN_CASE(DD)                                                                       \
    case DD:                                                                                       \
        return launch_attn<DD>(qh, kh, vh, kv_rows, oh, ws, B, H, Lq, n_groups, M, group_rows, scale, \
                               diag_bias, q_ld, kv_ld, st);
    switch (D) {
        FRESCO_ATTN_CASE(8)
        FRESCO_ATTN_CASE(16)
        FRESCO_ATTN_CASE(32)
        FRESCO_ATTN_CASE(40)
        FRESCO_ATTN_CASE(64)
        FRESCO_ATTN_CASE(80)
        FRESCO_ATTN_CASE(96)
        FRESCO_ATTN_CASE(128)
        default:
            return FRESCO_EUNSUPPORTED;
    }
#undef FRESCO_ATTN_CASE
}

extern "C" int fresco_attn_fwd(const void* q, const void* k, const void* v, const int32_t* kv_rows,
                               void* out, void* workspace, size_t workspace_bytes, int B, int H,
                               int Lq, int D, int n_groups, int M, int64_t group_rows, float scale,
                               float diag_bias, void* stream) {
    return fresco_attn_fwd_ld(q, k, v, kv_rows, out, workspace, workspace_bytes, B, H, Lq, D, n_groups, M,
                              group_rows, scale, diag_bias, (int64_t)H * D, (int64_t)H * D, stream);
}
