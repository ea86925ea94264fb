// Fused linear projections of the attention processor (reference: attn.to_q / to_k / to_v / to_out[0],
// src/diffusion_hacked.py:201, 214-215, 260-261, 375):   out_j = x W_j^T (+ b_j),  j < nw <= 3.
//
// The reference issues one GEMM per projection; each re-reads x and, for M = B*HW rows against 320 or
// 640 features, sits far below both rooflines in the library GEMM (37 us per 65536 x 320 x 320 product =
// 2.3 TB/s, 360 TFLOP/s).  Here x is read ONCE: a wave keeps its 32 rows of x resident in registers as
// MFMA B fragments (K/16 fragments of 8 halfs) and streams tiles of 32 weight rows through LDS -- the same
// swapped orientation as the flash kernel (S^T = W_tile x^T: one lane owns one row of x and ends up with
// 2 x 8 consecutive output features, i.e. two 16-byte stores per tile).  Weight tiles are DMA'd
// global -> LDS (global_load_lds_dwordx4), double buffered, rows padded to an odd multiple of 16 B
// (conflict-free ds_read_b128); the weights (<= 2.4 MB) stay L2-resident.
//
// grid (ceil(M/128), splits): 4 waves x 32 rows; blockIdx.y takes a contiguous range of feature tiles
// (splits > 1 only when M/128 alone cannot fill the chip).
// Measured (MI355X, M = 65536, K = 320): q,k,v in one launch 57 us vs 109 us for three library GEMMs; to_out 30 vs
// 40 us.  K = 640, M = 16384: 57 vs 77 us fused, but 30 vs 25 us for a single projection (the caller keeps the
// library GEMM there).  Where the 57 us go: 13 us output stores (kernel without them: 45 us), the rest the
// DMA -> 20 MFMA -> barrier steps at ~36 % of the MFMA rate; a 3-deep LDS ring with counted vmcnt (slab two steps
// ahead in flight), 256-row workgroups (half the weight traffic) and pairing two tiles' stores into whole 128-byte
// lines were measured: equal / 20 % slower / 12 % slower.
#include "common.h"

namespace fresco {

template <int K, int NWV>
struct ProjCfg {
    static constexpr int KC = 320;              // K chunk staged per step (halfs)
    static constexpr int NKC = K / KC;          // steps per feature tile
    static constexpr int KS = KC / 16;          // MFMA k-steps per step
    static constexpr int NXF = K / 16;          // resident x fragments
    static constexpr int ROWB = KC * 2 + 16;    // LDS bytes per weight row (41 chunks of 16 B: odd)
    static constexpr int CPR = ROWB / 16;
    static constexpr int TF = 32;               // weight rows (output features) per tile
    static constexpr int NP = (TF * CPR + 63) / 64;  // 1 KiB DMA pieces per tile (the last one is partly pad)
    static constexpr int PW = (NP + NWV - 1) / NWV;  // pieces per wave and step
    static constexpr int BUFB = PW * NWV * 1024;
    static constexpr int LDS_BYTES = 2 * BUFB;
};

template <int K, int NWV>
__global__ __launch_bounds__(NWV * 64, (NWV == 4 ? (K <= 320 ? 3 : 2) : 1)) void linear_kernel(
    const half_t* __restrict__ x, int64_t x_ld, const half_t* __restrict__ W0, const half_t* __restrict__ W1,
    const half_t* __restrict__ W2, const half_t* __restrict__ b0, const half_t* __restrict__ b1,
    const half_t* __restrict__ b2, half_t* __restrict__ out0, half_t* __restrict__ out1, half_t* __restrict__ out2, int64_t ld0, int64_t ld1,
    int64_t ld2, int M, int N, int nF, int tiles_per_split) {
    using Cfg = ProjCfg<K, NWV>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int row = blockIdx.x * (NWV * 32) + wave * 32 + l31;
    const int ft0 = blockIdx.y * tiles_per_split;
    const int ft1 = min(nF, ft0 + tiles_per_split);
    if (ft0 >= ft1) return;

    // x fragments: B operand of lane (row l31, half hi).  The contraction index is permuted (the same way
    // for W below) so that a lane's fragments are CONSECUTIVE in memory: within a K chunk, half hi of the
    // wave takes k = hi*KC/2 + 8*ks .. +7 -- 320 contiguous bytes per lane and chunk instead of 16 out of
    // every 32.
    half8_t xf[Cfg::NXF];
    {
        const half_t* xp = x + (int64_t)(row < M ? row : M - 1) * x_ld + hi * (Cfg::KC / 2);
#pragma unroll
        for (int kc = 0; kc < Cfg::NKC; ++kc)
#pragma unroll
            for (int ks = 0; ks < Cfg::KS; ++ks)
                xf[kc * Cfg::KS + ks] = *reinterpret_cast<const half8_t*>(xp + kc * Cfg::KC + ks * 8);
    }
    // A-tile row (lane & 31) is fed with weight row swap_bits_2_3(lane & 31): the 16 accumulator registers
    // of a lane then cover features 8*hi + (0..7) and 16 + 8*hi + (0..7) of the tile
    const int frow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

    // DMA: per-lane byte offset of its chunk inside a (feature tile, K chunk) slab of W, computed once
    uint32_t dma_off[Cfg::PW];
#pragma unroll
    for (int i = 0; i < Cfg::PW; ++i) {
        const int c = (i * NWV + wave_s) * 64 + lane;
        const int r = c / Cfg::CPR, dc = c % Cfg::CPR;
        dma_off[i] = (r < Cfg::TF && dc < Cfg::CPR - 1) ? (uint32_t)(r * K * 2 + dc * 16) : 0u;
    }
    // feature tile ft belongs to projection ft / tiles_per_out: every projection keeps its own (live) weight
    // matrix, nothing is stacked or cached on the host side.  The tile's slab address is scalar arithmetic on
    // values the loop below carries (wave-uniform): no per-lane selects in front of the DMA.
    const int tiles_per_out = N / Cfg::TF;
    auto tile_base = [&](int ft) __attribute__((always_inline)) -> const char* {
        const int jw = __builtin_amdgcn_readfirstlane(ft / tiles_per_out);
        // (arithmetic instead of a three-way select: the compiler turns the select into a scratch-memory table)
        const int64_t d1 = reinterpret_cast<const char*>(W1) - reinterpret_cast<const char*>(W0);
        const int64_t d2 = reinterpret_cast<const char*>(W2) - reinterpret_cast<const char*>(W0);
        const int64_t dj = (int64_t)(jw == 1) * d1 + (int64_t)(jw == 2) * d2;
        return reinterpret_cast<const char*>(W0) + dj + (int64_t)(ft - jw * tiles_per_out) * Cfg::TF * K * 2;
    };
    auto stage = [&](const char* tbase, int kc, int buf) __attribute__((always_inline)) {
        const char* src = tbase + kc * Cfg::KC * 2;  // wave-uniform
        char* dst = smem + buf * Cfg::BUFB + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < Cfg::PW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dma_off[i]),
                                             (__attribute__((address_space(3))) void*)(dst + i * NWV * 1024), 16, 0, 0);
    };

    // output cursor of this split: which tensor, which column
    int j = ft0 / tiles_per_out;
    int col = (ft0 % tiles_per_out) * Cfg::TF;

    const char* tb_cur = tile_base(ft0);
    stage(tb_cur, 0, 0);
    __syncthreads();
    int buf = 0;
    for (int ft = ft0; ft < ft1; ++ft) {
        const char* tb_next = ft + 1 < ft1 ? tile_base(ft + 1) : tb_cur;
        floatx16 acc0, acc1;  // two independent accumulation chains (even / odd k-steps)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = 0.f;
            acc1[r] = 0.f;
        }
#pragma unroll
        for (int kc = 0; kc < Cfg::NKC; ++kc) {
            // next slab: the other buffer was released by the barrier that ended the previous step
            if (kc + 1 < Cfg::NKC)
                stage(tb_cur, kc + 1, buf ^ 1);
            else if (ft + 1 < ft1)
                stage(tb_next, 0, buf ^ 1);
            const char* wr = smem + buf * Cfg::BUFB + frow * Cfg::ROWB + hi * Cfg::KC;  // hi * (KC/2) halfs
#pragma unroll
            for (int ks = 0; ks < Cfg::KS; ks += 2) {
                const half8_t a0 = *reinterpret_cast<const half8_t*>(wr + ks * 16);
                const half8_t a1 = *reinterpret_cast<const half8_t*>(wr + ks * 16 + 16);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xf[kc * Cfg::KS + ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xf[kc * Cfg::KS + ks + 1], acc1, 0, 0, 0);
            }
            __syncthreads();  // also drains this wave's DMA pieces before the buffers swap
            buf ^= 1;
        }
        // epilogue of the feature tile: + bias, fp16, two 16-byte stores per lane
        half_t* op = (j == 0) ? out0 : (j == 1 ? out1 : out2);
        const int64_t ld = (j == 0) ? ld0 : (j == 1 ? ld1 : ld2);
        if (row < M) {
            half_t* o = op + (int64_t)row * ld + col + hi * 8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                half8_t w;
                float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const half_t* bias = b0;  // (same: no pointer table)
                if (j == 1) bias = b1;
                if (j == 2) bias = b2;
                if (bias) {
                    const half8_t b8 = *reinterpret_cast<const half8_t*>(bias + col + half * 16 + hi * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = (float)b8[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (half_t)(acc0[half * 8 + e] + acc1[half * 8 + e] + bv[e]);
                *reinterpret_cast<half8_t*>(o + half * 16) = w;
            }
        }
        tb_cur = tb_next;
        col += Cfg::TF;
        if (col == N) {
            col = 0;
            ++j;
        }
    }
}

template <int K, int NWV>
static int launch_linear(const half_t* x, int64_t x_ld, const half_t* const* W, const half_t* const* bias,
                         half_t* out0, half_t* out1, half_t* out2, int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M,
                         int N, hipStream_t st) {
    using Cfg = ProjCfg<K, NWV>;
    // (per device and cheap: set on every launch rather than cached in a process-global flag)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_kernel<K, NWV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    const int nF = nw * N / Cfg::TF;
    const int row_blocks = (M + NWV * 32 - 1) / (NWV * 32);
    // enough workgroups for two rounds of the 256 CUs; every extra split re-reads x once
    int splits = (512 + row_blocks - 1) / row_blocks;
    if (splits > nF) splits = nF;
    if (splits < 1) splits = 1;
    const int tiles_per_split = (nF + splits - 1) / splits;
    splits = (nF + tiles_per_split - 1) / tiles_per_split;
    ProfScope ps(FRESCO_PROF_LINEAR, M, N, K, nw, st);
    hipLaunchKernelGGL((linear_kernel<K, NWV>), dim3(row_blocks, splits), dim3(NWV * 64), Cfg::LDS_BYTES, st, x, x_ld, W[0], W[1],
                       W[2], bias[0], bias[1], bias[2], out0, out1, out2, ld0, ld1, ld2, M, N, nF, tiles_per_split);
    return check_launch();
}

}  // namespace fresco

using namespace fresco;

extern "C" int fresco_linear(const void* x, int64_t x_ld, const void* W0, const void* W1, const void* W2,
                             const void* b0, const void* b1, const void* b2, void* out0, void* out1, void* out2,
                             int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M, int N, int K, void* stream) {
    if (!x || !W0 || !out0 || nw < 1 || nw > 3 || M <= 0 || N <= 0 || K <= 0) return FRESCO_EINVAL;
    if ((nw > 1 && (!out1 || !W1)) || (nw > 2 && (!out2 || !W2))) return FRESCO_EINVAL;
    if (x_ld < K || x_ld % 8 != 0) return FRESCO_EINVAL;
    if (ld0 < N || ld0 % 8 != 0 || (nw > 1 && (ld1 < N || ld1 % 8 != 0)) || (nw > 2 && (ld2 < N || ld2 % 8 != 0)))
        return FRESCO_EINVAL;
    if (N % 32 != 0 || (K != 320 && K != 640)) return FRESCO_EUNSUPPORTED;
    if ((int64_t)(M + 127) / 128 > 0x7fffffff) return FRESCO_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const half_t* xh = static_cast<const half_t*>(x);
    const half_t* wh[3] = {static_cast<const half_t*>(W0), static_cast<const half_t*>(W1), static_cast<const half_t*>(W2)};
    const half_t* bh[3] = {static_cast<const half_t*>(b0), static_cast<const half_t*>(b1), static_cast<const half_t*>(b2)};
    half_t* o0 = static_cast<half_t*>(out0);
    half_t* o1 = static_cast<half_t*>(out1);
    half_t* o2 = static_cast<half_t*>(out2);
    // 4 waves = 128 rows per workgroup (3 workgroups per CU); 256-row workgroups measured 20 % slower
    if (K == 320) return launch_linear<320, 4>(xh, x_ld, wh, bh, o0, o1, o2, ld0, ld1, ld2, nw, M, N, st);
    return launch_linear<640, 4>(xh, x_ld, wh, bh, o0, o1, o2, ld0, ld1, ld2, nw, M, N, st);
}
