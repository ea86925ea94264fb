// The flow network's dense layers: GMFlow's CNN encoder, transformer projections / FFN / LayerNorms and
// upsampler head (reference: src/ebsynth/deps/gmflow/gmflow/backbone.py:7-117, transformer.py:111-237, gmflow.py:44-90),
// which rounds 1-4 left to MIOpen / rocBLAS (85 % of the forward).  Everything is fp32 in and fp32 out -- the flows feed
// integer decisions (occlusion thresholds, pixel correspondences) -- but the products run on the fp16 matrix pipe the way
// attn32.hip's do: every fp32 operand x is split x = xh + xl into two halfs and a b is taken as ah bh + ah bl + al bh
// (three v_mfma_f32_32x32x16_f16 per 16 contraction steps; the dropped al bl term is < 2^-22 relative; fp32 accumulation).
//
// Data layout: activations are NHWC = (rows m = (image, y, x), channels) row-major, which IS the token layout (B, L, C) of
// the transformer -- no transposes between the encoder and the attention layers.  A tensor that feeds a product exists as a
// PAIR of fp16 planes (hi, lo) written by its producer (fn_prep_kernel / fn_layernorm_kernel / the GEMM epilogue), so the
// GEMM's loaders move 16-byte pieces of ready operands and no conversion sits between the loads and the MFMAs.
//
//   fn_gemm_kernel<BN>   out[m][n] = sum_k A(m, k) W[n][k] (+ bias, GELU / ReLU), A either row-major (linear layers) or the
//                        implicit im2col view of an NHWC tensor (k = (ky, kx, ci), zero padding): 256 x BN x 32 tiles,
//                        8 waves, operands by LDS-DMA into a 3-slot ring (source-side swizzle, counted vmcnt, one barrier
//                        per chunk), fp32 result and / or its (hi, lo) split and / or InstanceNorm partial sums written
//                        by the epilogue
//   fn_colstats_kernel   InstanceNorm2d statistics: mean and 1 / sqrt(biased var + eps) per (image, channel), fp64 partial
//                        sums in a fixed order (deterministic)
//   fn_prep_kernel       y = [relu]( [relu]((x - mean) rstd) + residual ) -> fp32 and / or (hi, lo), channel-padded rows
//   fn_layernorm_kernel  LayerNorm over 128 channels (+ residual), one wave per row
//   fn_conv7_rgb_kernel  the 7 x 7 / stride 2 stem on 3 input channels: direct fp32 FMAs (K = 147 is no MFMA shape)
#include "common.h"

namespace fresco {

// x * scale = h + l.  The matrix pipe FLUSHES fp16 subnormals (attn32.hip), so a lo piece below 6.1e-5 would be lost and the
// operand would be no better than fp16: every plane is written pre-scaled by a power of two -- activations by 2^6 (lo pieces
// stay normal down to |x| = 2e-3, values up to 1000 fit), weights by 2^10 (|w| from 1.2e-4 to 60) -- and the GEMM scales its
// fp32 accumulators back exactly.  Beyond the range the scaled value saturates (finite, wrong) instead of becoming inf / NaN
// -- and the producer says so: fn_split returns true for a value it had to clamp (or a NaN), every producer kernel ORs
// that into the caller's `range_flag` word (fresco_fn_prep / _layernorm / _gemm), and the host side of the flow network
// re-runs the forward with library ops when the word is set (fresco_amd/gmflow.py) -- the flows feed integer decisions.
__device__ __forceinline__ bool fn_split(float x, float scale, half_t& h, half_t& l) {
    const float xs = x * scale;
    x = fminf(fmaxf(xs, -65000.f), 65000.f);
    h = (half_t)x;
    l = (half_t)(x - (float)h);
    return !(fabsf(xs) <= 65000.f);
}
__device__ __forceinline__ void fn_flag_range(int32_t* range_flag, bool sat) {
    if (range_flag && sat) atomicOr(range_flag, 1);  // (rare path)
}

// ------------------------------------------------------------------------------------------------------------------------
struct FnConv {  // implicit-GEMM view of an NHWC tensor; kh == 0: A is a plain row-major matrix
    int kh, kw, stride, pad, H, W, OH, OW, cin;  // cin = channels per pixel of the (padded) input rows
    int tiled;                                   // 1: a workgroup's 256 rows are a 16 x 16 patch of output pixels
};

constexpr int FN_BM = 256, FN_BK = 32, FN_NS = 3;

// Row r (0 .. 255) of row block `blk` -> output row m.  Linear layers and odd-sized maps: m = blk * 256 + r.  Convolutions on
// maps of whole 16 x 16 patches: the block is a PATCH of one image, so that the 3 x 3 taps of its 256 pixels touch an
// 18 x 18 window of the input (the nine (ky, kx) chunks re-read it from L1 / L2) instead of a 256-pixel row segment times
// three rows.  The output keeps its NHWC row order either way.
__device__ __forceinline__ int fn_row_of(const FnConv& cv, int blk, int r) {
    if (!cv.tiled) return blk * FN_BM + r;
    const int tw = cv.OW >> 4, per_img = (cv.OH >> 4) * tw;
    const int img = blk / per_img, t = blk - img * per_img;
    const int ty = t / tw, tx = t - ty * tw;
    return (img * cv.OH + ty * 16 + (r >> 4)) * cv.OW + tx * 16 + (r & 15);
}

// one LDS-DMA piece: 64 lanes x 16 bytes, lane l's bytes land at lds_dst + 16 l (M0 is written in the statement that uses it)
__device__ __forceinline__ void fn_dma16(const void* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");  // (m0 cannot be listed as a clobber: hipcc rejects it as a reserved register; it does not keep values in m0 across statements on gfx9+)
}
template <int N_>
__device__ __forceinline__ void fn_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N_) : "memory");
}

// 256 x BN x 32 tiles, 8 waves as 4 x 2 (wave tiles 64 x BN/2), one workgroup per CU.  Operands arrive by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of three slots, chunks two steps ahead, behind
// counted vmcnt waits and ONE barrier per chunk (the protocol of opt_fast.hip's Gram / S V kernels).  A slot holds the four
// planes [A hi | A lo | W hi | W lo] as unpadded 64-byte rows (32 halfs): the DMA writes lane-linear, so the bank swizzle is on
// the SOURCE side -- the lane that owns LDS position q of row R fetches 16-byte piece q ^ ((R >> 2) & 3) of that row -- and
// on the read side (fragment piece p of row R sits at position p ^ ((R >> 2) & 3)): conflict-free ds_read_b128 in the b128
// lane groups.  Rows outside the problem (m >= M, n >= N, the zero padding of a convolution) are fetched from a 16-byte
// page of zeros.
//
// PATCH (3 x 3, stride 1, pad 1 convolutions on maps of whole 16 x 16 patches): the im2col view re-reads every input pixel
// nine times -- 32 KB of A pieces per 32-wide K chunk, and with a 3-slot ring only two chunks in flight per CU, the loop
// runs at the DMA's latency (13 B/clk/CU at 256^2), not at the matrix pipe's rate.  This form turns the K loop inside out
// -- channel chunks OUTER, the nine taps INNER -- and keeps the 18 x 18 input window of the patch, 32 channels deep, in
// LDS: 324 pixel rows of 64 bytes per plane, fetched ONCE per channel chunk (two slots: chunk c + 1 arrives, one piece per
// tap step, under the taps of chunk c), the A fragments of tap (ky, kx) are ds_read_b128 at pixel (oy + ky, ox + kx).
// Only the weights still stream per step (their own 3-slot ring): 4.6 + 8 KB per step instead of 32 + 8.  Window pixel
// P = 18 py + px keeps its 16-byte pieces at position piece ^ ((py + 2 (P >> 2)) & 3): conflict-free for all nine taps
// in the ds_read_b128 lane groups (searched by enumeration; (P >> 2) & 3 alone is 2-way on four of the taps).
template <int BN, bool PATCH>
__global__ __launch_bounds__(512, 1) void fn_gemm_kernel(const half_t* __restrict__ a_hi, const half_t* __restrict__ a_lo,
                                                         int64_t lda, FnConv cv, const half_t* __restrict__ w_hi,
                                                         const half_t* __restrict__ w_lo, const float* __restrict__ bias,
                                                         float* __restrict__ out, half_t* __restrict__ o_hi,
                                                         half_t* __restrict__ o_lo, int64_t ldc, int64_t ldo, int M, int N,
                                                         int K, int act, float acc_scale, float split_scale,
                                                         double* __restrict__ stats, const void* __restrict__ zeros,
                                                         const int32_t* __restrict__ a_rows,
                                                         const int32_t* __restrict__ out_rows, int32_t* range_flag,
                                                         int out_cb, int64_t out_bs, int nb_xmap) {
    // out_cb > 0 (fp32 output only): the N columns are out_cb-wide BLOCKS that go to separate (M, out_cb) matrices out_bs
    // floats apart (row stride ldc) -- q | k | v of one source in ONE product, each landing in a matrix of its own
    // a_rows / out_rows (linear layers; NULL = identity): problem row m reads input row a_rows[m] and its results go to output
    // row out_rows[m] -- the token gather / scatter of the (shifted-)window attention folded into the projections around it
    constexpr int BM = FN_BM, BK = FN_BK, NS = FN_NS;
    constexpr int A_PL = BM * 64, W_PL = BN * 64;          // bytes per plane and slot
    constexpr int SLOT = 2 * A_PL + 2 * W_PL;
    constexpr int WN = BN / 2, NB = WN / 32;
    constexpr int NPW = 4 + (BN == 128 ? 2 : 1);           // DMA pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> (row block bx, column block by).  The grid is 1-D and the hardware deals consecutive workgroups to the 8
    // XCDs in turn, each with an L2 of its own: with nb > 1 column blocks, XCD x takes row blocks x, x + 8, .. and walks
    // the nb column blocks of one row block back to back, so the 256 x K operand rows are fetched into ONE L2 once and hit
    // there nb - 1 times (column block outermost, the plain order, re-reads the whole A from HBM / MALL nb times: at
    // (65536, 1024, 256) 536 MB instead of 67).  The last rb % 8 row blocks keep the plain order.
    int bx, by;
    {
        // (nb_xmap = nb | xmap << 16; rb = the row blocks of the problem)
        const int nb = nb_xmap & 0xffff, xmap = nb_xmap >> 16, rb = (M + FN_BM - 1) / FN_BM;
        const int L = blockIdx.x, full = (rb >> 3) * 8 * nb;
        if (nb == 1) {
            bx = L;
            by = 0;
        } else if (!xmap) {
            by = L / rb;
            bx = L - by * rb;
        } else if (L < full) {
            const int j = L >> 3;
            bx = (j / nb) * 8 + (L & 7);
            by = j % nb;
        } else {
            const int t = L - full;
            bx = (rb >> 3) * 8 + t / nb;
            by = t % nb;
        }
    }
    const int n0 = by * BN;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);

    const int q = lane & 3;
    // W: BN = 128: piece `wave` of both W planes; BN = 64: waves 0-3 piece `wave` of W hi, waves 4-7 piece wave - 4 of W lo
    const int w_piece = BN == 128 ? wave : (wave & 3);
    const int w_R = w_piece * 16 + (lane >> 2);
    const int w_pc = q ^ ((w_R >> 2) & 3);
    const bool w_ok = n0 + w_R < N;
    const int64_t w_off = (int64_t)(w_ok ? n0 + w_R : 0) * K + w_pc * 8;

    floatx16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l31 >> 2) & 3;
    if constexpr (!PATCH) {
        // ---- DMA sources.  A: this wave copies pieces 2 wave, 2 wave + 1 of both A planes (16 rows each); a lane owns LDS
        // position q = lane & 3 of row R = 16 piece + (lane >> 2) and fetches source piece q ^ ((R >> 2) & 3) of that row
        int a_pc[2], a_iy[2], a_ix[2];
        int64_t a_base[2];
        bool a_ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int R = (2 * wave + i) * 16 + (lane >> 2);
            a_pc[i] = q ^ ((R >> 2) & 3);
            const int m = fn_row_of(cv, bx, R);
            a_ok[i] = m < M;
            const int mm = a_ok[i] ? m : 0;
            if (cv.kh == 0) {
                a_base[i] = (int64_t)(a_rows ? a_rows[mm] : mm) * lda;
                a_iy[i] = a_ix[i] = 0;
            } else {
                const int ohw = cv.OH * cv.OW;
                const int img = mm / ohw, r = mm - img * ohw;
                const int oy = r / cv.OW, ox = r - oy * cv.OW;
                a_iy[i] = oy * cv.stride - cv.pad;
                a_ix[i] = ox * cv.stride - cv.pad;
                a_base[i] = (int64_t)img * cv.H * cv.W;
            }
        }
        const int nk = K / BK;
        auto stage = [&](int kc, int slot) __attribute__((always_inline)) {
            const int k0 = kc * BK;
            int ky = 0, kx = 0, c0 = k0;
            if (cv.kh != 0) {  // a 32-channel chunk lies inside one (ky, kx): cin % 32 == 0
                const int t = k0 / cv.cin;
                c0 = k0 - t * cv.cin;
                ky = t / cv.kw;
                kx = t - ky * cv.kw;
            }
            const uint32_t sb = lds0 + (uint32_t)(slot * SLOT);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bool ok = a_ok[i];
                int64_t off;
                if (cv.kh == 0) {
                    off = a_base[i] + k0 + a_pc[i] * 8;
                } else {
                    const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                    ok = ok && iy >= 0 && iy < cv.H && ix >= 0 && ix < cv.W;
                    off = (a_base[i] + (int64_t)iy * cv.W + ix) * lda + c0 + a_pc[i] * 8;
                }
                const void* ph = ok ? static_cast<const void*>(a_hi + off) : zeros;
                const void* pl = ok ? static_cast<const void*>(a_lo + off) : zeros;
                fn_dma16(ph, sb + (uint32_t)((2 * wave + i) * 1024));
                fn_dma16(pl, sb + (uint32_t)(A_PL + (2 * wave + i) * 1024));
            }
            if (BN == 128) {
                fn_dma16(w_ok ? static_cast<const void*>(w_hi + w_off + k0) : zeros, sb + (uint32_t)(2 * A_PL + wave * 1024));
                fn_dma16(w_ok ? static_cast<const void*>(w_lo + w_off + k0) : zeros, sb + (uint32_t)(2 * A_PL + W_PL + wave * 1024));
            } else {
                const half_t* wp = wave < 4 ? w_hi : w_lo;
                fn_dma16(w_ok ? static_cast<const void*>(wp + w_off + k0) : zeros,
                         sb + (uint32_t)(2 * A_PL + (wave < 4 ? 0 : W_PL) + (wave & 3) * 1024));
            }
        };

        // (Round 6, measured and not kept: fragments of chunk kc + 1 read into a second register set under the MFMAs of chunk
        // kc, three chunks in flight -- bit-identical, no faster (EXPERIMENTS.md 4): tools/ubench_fn_gemm_k.py puts a
        // 256 x 128 x 32 chunk at 2.0 us on a full chip = 48 KB per CU and chunk at 10 B/clk/CU, the rate at which a CU's
        // LDS-DMA requests are served when all 256 stream, not the LDS-read / MFMA lockstep behind the barrier.)
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int slot = 0;
        for (int kc = 0; kc < nk; ++kc) {
            // own pieces of chunk kc have landed (what may still fly: chunk kc + 1's), then everyone's, and every wave is done
            // with chunk kc - 1: its slot takes chunk kc + 2
            if (kc + 1 < nk)
                fn_wait_barrier<NPW>();
            else
                fn_wait_barrier<0>();
            if (kc + 2 < nk) stage(kc + 2, slot >= 1 ? slot - 1 : NS - 1);
            const char* s = smem + slot * SLOT;
            const char* sa = s + (wm * 64 + l31) * 64;
            const char* sw_ = s + 2 * A_PL + (wn * WN + l31) * 64;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int pos = ((ks * 2 + hi) ^ sw) * 16;
                half8_t ah[2], al[2], bh[NB], bl[NB];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const half8_t*>(sa + i * 32 * 64 + pos);
                    al[i] = *reinterpret_cast<const half8_t*>(sa + A_PL + i * 32 * 64 + pos);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    bh[j] = *reinterpret_cast<const half8_t*>(sw_ + j * 32 * 64 + pos);
                    bl[j] = *reinterpret_cast<const half8_t*>(sw_ + W_PL + j * 32 * 64 + pos);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            slot = slot == NS - 1 ? 0 : slot + 1;
        }
    } else {
        // ---- the window of this patch: block -> (image, patch row, patch column); this wave copies pieces wave, wave + 8,
        // wave + 16 of both planes (24 pieces of 16 pixel rows per plane, rows >= 324 and pixels outside the image from the
        // zero page); a lane owns LDS position q of pixel row R and fetches source piece q ^ swizzle(R)
        constexpr int P_PL = 24 * 1024, P_SLOT = 2 * P_PL, W_SLOT = 2 * W_PL;
        constexpr int NW = BN == 128 ? 2 : 1;
        const int tw = cv.OW >> 4, per_img = (cv.OH >> 4) * tw;
        const int img = bx / per_img, tt = bx - img * per_img;
        const int ty = tt / tw, tx = tt - ty * tw;
        int64_t p_off[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int R = (wave + 8 * n) * 16 + (lane >> 2);
            const int py = R / 18, px = R - py * 18;
            const int iy = ty * 16 - 1 + py, ix = tx * 16 - 1 + px;
            const bool ok = R < 324 && iy >= 0 && iy < cv.H && ix >= 0 && ix < cv.W;
            p_off[n] = ok ? (((int64_t)img * cv.H + iy) * cv.W + ix) * lda + ((q ^ ((py + 2 * (R >> 2)) & 3)) * 8) : -1;
        }
        const int nc = cv.cin >> 5, total = nc * 9;
        // output pixel of this lane's A rows: (oy, ox) = (4 wm + 2 i + (l31 >> 4), l31 & 15)
        const int oyb = wm * 4 + (l31 >> 4), ox = l31 & 15;
#define FN_PATCH_PIECE(C_, N_)                                                                                              \
    do {                                                                                                                    \
        const int64_t po_ = p_off[(N_) % 3];                                                                                \
        const half_t* pl_ = (N_) < 3 ? a_hi : a_lo;                                                                         \
        fn_dma16(po_ >= 0 ? static_cast<const void*>(pl_ + po_ + (C_) * 32) : zeros,                                        \
                 lds0 + (uint32_t)((((C_) & 1) * P_SLOT) + ((N_) / 3) * P_PL + (wave + 8 * ((N_) % 3)) * 1024));            \
    } while (0)
#define FN_PATCH_WSTAGE(K0_, SLOT_)                                                                                         \
    do {                                                                                                                    \
        const uint32_t sb_ = lds0 + (uint32_t)(2 * P_SLOT + (SLOT_) * W_SLOT);                                              \
        if (BN == 128) {                                                                                                    \
            fn_dma16(w_ok ? static_cast<const void*>(w_hi + w_off + (K0_)) : zeros, sb_ + (uint32_t)(wave * 1024));         \
            fn_dma16(w_ok ? static_cast<const void*>(w_lo + w_off + (K0_)) : zeros, sb_ + (uint32_t)(W_PL + wave * 1024));  \
        } else {                                                                                                            \
            const half_t* wp_ = wave < 4 ? w_hi : w_lo;                                                                     \
            fn_dma16(w_ok ? static_cast<const void*>(wp_ + w_off + (K0_)) : zeros,                                          \
                     sb_ + (uint32_t)((wave < 4 ? 0 : W_PL) + (wave & 3) * 1024));                                          \
        }                                                                                                                   \
    } while (0)
        // one tap step: W chunk (tap T_, channels 32 c ..) against the window pixels (oy + ky, ox + kx).  Issue order per step:
        // the W pieces of step s + 2, then ONE window piece of chunk c + 1 (taps 0-5) -- so the pieces younger than step s's
        // W pieces at its wait are step s + 1's W pieces plus the window pieces of steps s - 2 and s - 1 of the same chunk
#define FN_PATCH_STEP(T_)                                                                                                   \
    do {                                                                                                                    \
        constexpr int older_ = (((T_) >= 1 && (T_) <= 6) ? 1 : 0) + (((T_) >= 2 && (T_) <= 7) ? 1 : 0);                     \
        const int s_ = c * 9 + (T_);                                                                                        \
        if (s_ + 1 >= total)                                                                                                \
            fn_wait_barrier<0>();                                                                                           \
        else if (nxt)                                                                                                       \
            fn_wait_barrier<NW + older_>();                                                                                 \
        else                                                                                                                \
            fn_wait_barrier<NW>();                                                                                          \
        if (s_ + 2 < total) {                                                                                               \
            const int t2_ = (T_) + 2 >= 9 ? (T_) + 2 - 9 : (T_) + 2, c2_ = (T_) + 2 >= 9 ? c + 1 : c;                       \
            FN_PATCH_WSTAGE(t2_ * cv.cin + c2_ * 32, slot >= 1 ? slot - 1 : NS - 1);                                        \
        }                                                                                                                   \
        if ((T_) < 6 && nxt) FN_PATCH_PIECE(c + 1, (T_));                                                                   \
        const char* pa_ = smem + (c & 1) * P_SLOT;                                                                          \
        const char* sw2_ = smem + 2 * P_SLOT + slot * W_SLOT + (wn * WN + l31) * 64;                                        \
        int aoff_[2];                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                     \
            const int py_ = oyb + 2 * i + (T_) / 3, P_ = py_ * 18 + ox + (T_) % 3;                                          \
            aoff_[i] = P_ * 64 + ((hi ^ ((py_ + 2 * (P_ >> 2)) & 3)) * 16);                                                 \
        }                                                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                                            \
            const int pos = ((ks * 2 + hi) ^ sw) * 16;                                                                      \
            half8_t ah[2], al[2], bh[NB], bl[NB];                                                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
                ah[i] = *reinterpret_cast<const half8_t*>(pa_ + (aoff_[i] ^ (ks * 32)));                                    \
                al[i] = *reinterpret_cast<const half8_t*>(pa_ + P_PL + (aoff_[i] ^ (ks * 32)));                             \
            }                                                                                                               \
            _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                                \
                bh[j] = *reinterpret_cast<const half8_t*>(sw2_ + j * 32 * 64 + pos);                                        \
                bl[j] = *reinterpret_cast<const half8_t*>(sw2_ + W_PL + j * 32 * 64 + pos);                                 \
            }                                                                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
                _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                            \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);                   \
                }                                                                                                           \
        }                                                                                                                   \
        slot = slot == NS - 1 ? 0 : slot + 1;                                                                               \
    } while (0)
#pragma unroll
        for (int n = 0; n < 6; ++n) FN_PATCH_PIECE(0, n);
        FN_PATCH_WSTAGE(0, 0);
        FN_PATCH_WSTAGE(cv.cin, 1);
        int slot = 0;
        for (int c = 0; c < nc; ++c) {
            const bool nxt = c + 1 < nc;
            FN_PATCH_STEP(0);
            FN_PATCH_STEP(1);
            FN_PATCH_STEP(2);
            FN_PATCH_STEP(3);
            FN_PATCH_STEP(4);
            FN_PATCH_STEP(5);
            FN_PATCH_STEP(6);
            FN_PATCH_STEP(7);
            FN_PATCH_STEP(8);
        }
#undef FN_PATCH_STEP
#undef FN_PATCH_WSTAGE
#undef FN_PATCH_PIECE
    }
    // ---- epilogue: lane (l31, hi) holds column n = .. + l31 and rows (r & 3) + 8 (r >> 2) + 4 hi of each 32 x 32 block.
    // fp32 results go out as they sit (a store instruction writes 32 consecutive floats of one row = one 128-byte line).
    // The (hi, lo) planes would be 2-byte stores, 64 bytes per row and instruction: each plane is transposed through the
    // wave's share of the (now free) ring instead -- [64 rows][WN columns] halfs -- and leaves as 16-byte pieces.
    // stats != NULL: per-column sum and sum of squares of the results over this wave's 64 rows, fp64, into slab
    // (row block * 4 + wm) of the InstanceNorm partial-sum buffer (fn_colstats_final_kernel adds the slabs in order).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the ring
    constexpr int TROW = WN * 2 + 16;  // bytes per transposed row (odd number of 16-byte units)
    char* tp = smem + wave * (64 * TROW);
    float bcol[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        bcol[j] = (bias && n < N) ? bias[n] : 0.f;
    }
    // The output row of every accumulator register, BEFORE the stores (-1: a row outside the problem).  With the out_rows
    // table looked up inside the store loop hipcc put an s_waitcnt vmcnt(0) in front of every store -- on this target it
    // counts stores as well, so the 32 x NB stores of a lane left one at a time, each behind the previous one's round trip:
    // ~19 us of a 27 us workgroup at (65536, 128, 128), with or without a table (round 6, tools/ubench_fn_gemm.py).
    int orow[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = fn_row_of(cv, bx, wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
            orow[i][r] = m < M ? ((out && out_rows) ? out_rows[m] : m) : -1;
        }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        const bool nok = n < N;
        // this lane's output column: matrix n / out_cb, column n % out_cb when the product is split into column blocks
        float* ocol = out;
        if (out) ocol += out_cb > 0 ? (int64_t)(n / out_cb) * out_bs + (n % out_cb) : (int64_t)n;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * acc_scale + bcol[j];
                if (act == 1) v = fmaxf(v, 0.f);
                if (act == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));  // nn.GELU() (exact, erf form)
                acc[i][j][r] = v;
            }
        // (one branch per column block around the stores, a row test per store only where rows can lie outside the problem:
        // a patch form's 256 rows never do)
        if (out && nok) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (PATCH || orow[i][r] >= 0) ocol[(int64_t)orow[i][r] * ldc] = acc[i][j][r];
        }
        double s1 = 0.0, s2 = 0.0;
        if (stats) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double v = (PATCH || orow[i][r] >= 0) ? (double)acc[i][j][r] : 0.0;
                    s1 += v;
                    s2 += v * v;
                }
        }
        if (stats) {  // the two half-waves hold the two row halves of the same column: add them in a fixed order
            const double t1 = __shfl_xor(s1, 32, 64), t2 = __shfl_xor(s2, 32, 64);
            if (hi == 0 && nok) {
                double* o = stats + (((int64_t)bx * 4 + wm) * N + n) * 2;
                o[0] = s1 + t1;
                o[1] = s2 + t2;
            }
        }
    }
    if (o_hi) {
        constexpr int PPR = WN / 8;  // 16-byte pieces per row
        bool sat = false;
        int prow[PPR];  // (as orow above: the rows of this lane's transposed pieces, looked up before any store)
#pragma unroll
        for (int it = 0; it < PPR; ++it) {
            const int m = fn_row_of(cv, bx, wm * 64 + (it * 64 + lane) / PPR);
            prow[it] = m < M ? (out_rows ? out_rows[m] : m) : -1;
        }
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        half_t h, l;
                        sat |= fn_split(acc[i][j][r], split_scale, h, l);
                        const int off = (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * TROW + (j * 32 + l31) * 2;
                        *reinterpret_cast<half_t*>(tp + off) = plane == 0 ? h : l;
                    }
            // (the wave that wrote is the wave that reads: the LDS executes a wave's operations in order)
            half_t* op = plane == 0 ? o_hi : o_lo;
#pragma unroll
            for (int it = 0; it < PPR; ++it) {
                const int id = it * 64 + lane, rr = id / PPR, pc = id % PPR;
                const int n = n0 + wn * WN + pc * 8;
                const half8_t v8 = *reinterpret_cast<const half8_t*>(tp + rr * TROW + pc * 16);
                if (prow[it] >= 0 && n < N)  // (N % 8 == 0: launcher)
                    *reinterpret_cast<half8_t*>(op + (int64_t)prow[it] * ldo + n) = v8;
            }
        }
        // (accumulators of rows / columns outside the problem are products with the zero page: never out of range)
        fn_flag_range(range_flag, sat);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// InstanceNorm2d statistics of x (n_img * rows, C) fp32: pass 1 = fp64 partial sums over row slabs, pass 2 = fixed-order
// combine.  grid (slabs, n_img), 256 threads: thread t owns channel t % CP and every (256 / CP)-th row of the slab.
__global__ __launch_bounds__(256) void fn_colstats_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                                  int rows, int C, int slab_rows) {
    __shared__ double s1[256], s2[256];
    const int img = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    const int groups = 256 / C > 0 ? 256 / C : 1;
    const int r0 = slab * slab_rows, r1 = min(rows, r0 + slab_rows);
    for (int cb = 0; cb < C; cb += 256) {  // (C <= 256 in this network: one trip)
        const int c = cb + tid % (C < 256 ? C : 256), g = tid / (C < 256 ? C : 256);
        double a = 0.0, b = 0.0;
        if (g < groups && c < C) {
            const float* p = x + ((int64_t)img * rows) * C + c;
            for (int r = r0 + g; r < r1; r += groups) {
                const double v = (double)p[(int64_t)r * C];
                a += v;
                b += v * v;
            }
        }
        s1[tid] = a;
        s2[tid] = b;
        __syncthreads();
        if (g == 0 && c < C) {
            for (int gg = 1; gg < groups; ++gg) {
                a += s1[tid + gg * C];
                b += s2[tid + gg * C];
            }
            double* o = part + (((int64_t)img * gridDim.x + slab) * C + c) * 2;
            o[0] = a;
            o[1] = b;
        }
        __syncthreads();
    }
}

// Combine the slabs of partial sums: grid (ceil(C / 16), n_img), 256 threads = 16 channels x 16 slab lanes.  Lane g adds
// slabs g, g + 16, ... in order, thread g = 0 of a channel adds the 16 lane sums in order: a fixed summation order whatever
// the launch looks like.  (Round 5, first form: ONE thread per (image, channel) walking all slabs -- 1024 dependent loads at
// 256 x 256: 160 us average, 464 us worst, 12 % of the flow network's forward in profiles/r05_gmflow_kernel_stats.csv.)
__global__ __launch_bounds__(256) void fn_colstats_final_kernel(const double* __restrict__ part, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int slabs, int C, int rows,
                                                                float eps) {
    __shared__ double s1[16][17], s2[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl, img = blockIdx.y;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int s = g; s < slabs; s += 16) {
            const double* p = part + (((int64_t)img * slabs + s) * C + c) * 2;
            a += p[0];
            b += p[1];
        }
    s1[g][cl] = a;
    s2[g][cl] = b;
    __syncthreads();
    if (g == 0 && c < C) {
        a = 0.0;
        b = 0.0;
        for (int k = 0; k < 16; ++k) {
            a += s1[k][cl];
            b += s2[k][cl];
        }
        const double m = a / rows;
        double var = b / rows - m * m;  // biased variance (InstanceNorm2d)
        if (var < 0.0) var = 0.0;
        mean[(int64_t)img * C + c] = (float)m;
        rstd[(int64_t)img * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// y = relu_b?( relu_a?((x - mean) * rstd) + residual ), 4 channels per thread; writes fp32 y (ld C) and / or the (hi, lo)
// planes with row stride ldo >= C (channels C .. ldo-1 are zeroed: K padding of the next product)
__global__ __launch_bounds__(256) void fn_prep_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ res,
                                                      float* __restrict__ y, half_t* __restrict__ o_hi,
                                                      half_t* __restrict__ o_lo, int64_t M, int C, int ldo,
                                                      int rows_per_img, int relu_a, int relu_b, float split_scale,
                                                      int32_t* range_flag) {
    const int q = ldo / 4;  // quads per output row (C % 4 == 0, ldo % 4 == 0)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * q) return;
    const int64_t m = idx / q;
    const int c = (int)(idx - m * q) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const floatx4 t = *reinterpret_cast<const floatx4*>(x + m * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t[e];
        if (mean) {
            const int64_t s = (m / rows_per_img) * C + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean[s + e]) * rstd[s + e];
        }
        if (relu_a)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (res) {
            const floatx4 t2 = *reinterpret_cast<const floatx4*>(res + m * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t2[e];
        }
        if (relu_b)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (y) {
            floatx4 t3 = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<floatx4*>(y + m * C + c) = t3;
        }
    }
    if (o_hi) {
        half4_t h, l;
        bool sat = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            half_t hh, ll;
            sat |= fn_split(v[e], split_scale, hh, ll);
            h[e] = hh;
            l[e] = ll;
        }
        *reinterpret_cast<half4_t*>(o_hi + m * ldo + c) = h;
        *reinterpret_cast<half4_t*>(o_lo + m * ldo + c) = l;
        if (sat && range_flag) atomicOr(range_flag, 1);  // (rare; threads past M * q have returned: no wave-wide vote here)
    }
}

// LayerNorm over C = 128 channels (eps inside the sqrt, affine), one wave per row, two channels per lane:
// y = [res +] ((x - mean) / sqrt(var + eps)) * gamma + beta
__global__ __launch_bounds__(256) void fn_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           float* __restrict__ y, half_t* __restrict__ o_hi,
                                                           half_t* __restrict__ o_lo, int64_t ldy, int64_t ldo, int64_t M,
                                                           float eps, float split_scale, int32_t* range_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float a = x[m * 128 + lane * 2], b = x[m * 128 + lane * 2 + 1];
    const float mean = wave_sum(a + b) * (1.f / 128.f);
    const float da = a - mean, db = b - mean;
    const float var = wave_sum(da * da + db * db) * (1.f / 128.f);
    const float r = 1.f / sqrtf(var + eps);
    float ya = da * r * gamma[lane * 2] + beta[lane * 2];
    float yb = db * r * gamma[lane * 2 + 1] + beta[lane * 2 + 1];
    if (res) {
        ya += res[m * 128 + lane * 2];
        yb += res[m * 128 + lane * 2 + 1];
    }
    if (y) {
        y[m * ldy + lane * 2] = ya;
        y[m * ldy + lane * 2 + 1] = yb;
    }
    if (o_hi) {
        half_t h0, l0, h1, l1;
        bool sat = fn_split(ya, split_scale, h0, l0);
        sat |= fn_split(yb, split_scale, h1, l1);
        if (sat && range_flag) atomicOr(range_flag, 1);
        o_hi[m * ldo + lane * 2] = h0;
        o_hi[m * ldo + lane * 2 + 1] = h1;
        o_lo[m * ldo + lane * 2] = l0;
        o_lo[m * ldo + lane * 2 + 1] = l1;
    }
}

// The stem: Conv2d(3, 64, 7, stride 2, padding 3, bias=False) on NHWC fp32 (backbone.py:69).  Rounds 5: direct fp32 FMAs from
// an LDS patch (K = 147 "is no MFMA shape"): 496 us at 16 x 512^2, LDS-read bound (one broadcast b128 per four FMAs).
// Round 6: on the matrix pipe like every other product of the network -- the contraction index is laid out as
// k' = 32 ky + (3 kx + ci), 21 real positions per kernel row + 11 ZERO WEIGHTS, K' = 224 = 14 k-steps: for output pixel
// (oy, ox) the 32 values of kernel row ky are the 32 CONSECUTIVE floats of input row 2 oy - 3 + ky starting at pixel
// 2 ox - 3 (NHWC, 3 channels), so an A fragment is 8 consecutive values of the staged row -- no im2col, no gather; what the
// 11 surplus positions read (the next pixels of the row: finite) meets zero weights.
//   block = 8 waves = 2 output rows x 256 output columns, wave = 64 pixels x 64 channels; LDS: the 9 input rows of the tile
//   as (hi, lo) fp16 planes (split once per input value at staging, 2^6 pre-scale as every activation plane) + the weight
//   planes (64 rows of 224 halfs, 464-byte rows = an odd number of 16-byte units: conflict-free b128 rows); per k-step and
//   wave 16 ds_read_b32 (A: 4-byte aligned runs) + 4 ds_read_b128 (W) feed 12 MFMAs; epilogue: fp32 NHWC rows (a store =
//   32 consecutive floats) and, stats != NULL, the InstanceNorm partial sums of the wave's 64 pixels (slab layout of
//   fn_gemm_kernel's epilogue: fresco_fn_colstats_finish combines them).
constexpr int C7_PROW = 1568;  // halfs per staged input row: (2 * 256 + 5) * 3 = 1551 values, reads run to 6 * 255 + 31
constexpr int C7_ROWS = 9;     // input rows of two output rows
constexpr int C7_K = 224, C7_WROW = 464;
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
static_assert(C7_K == 28 * 8 && C7_WROW >= C7_K * 2 && (C7_WROW / 16) % 2 == 1, "stem weight rows: 28 pieces, odd LDS pitch");
__global__ __launch_bounds__(512, 1) void fn_conv7_rgb_kernel(const float* __restrict__ x, const half_t* __restrict__ w_hi,
                                                              const half_t* __restrict__ w_lo, float* __restrict__ out,
                                                              double* __restrict__ stats, int H, int W, int OH, int OW,
                                                              float in_scale, float acc_scale, int32_t* range_flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* xh = reinterpret_cast<half_t*>(smem);
    half_t* xl = xh + C7_ROWS * C7_PROW;
    char* wh = smem + 2 * C7_ROWS * C7_PROW * 2;
    char* wl = wh + 64 * C7_WROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int img = blockIdx.z, oy0 = blockIdx.y * 2, ox0 = blockIdx.x * 256;
    // ---- weights: 64 rows x 28 16-byte pieces per plane = 7 pieces per thread (all loads, then all LDS writes)
    {
        uintx4 wv[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + 512 * k, pl = i / (64 * 28), r = i - pl * (64 * 28);
            wv[k] = *reinterpret_cast<const uintx4*>((pl ? w_lo : w_hi) + r * 8);  // (piece pc of row n: n * 224 + pc * 8 = r * 8)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = tid + 512 * k, pl = i / (64 * 28), r = i - pl * (64 * 28);
            const int n = r / 28, pc = r - n * 28;
            *reinterpret_cast<uintx4*>((pl ? wl : wh) + n * C7_WROW + pc * 16) = wv[k];
        }
    }
    // ---- input rows: value e = 3 cx + ci of staged row ry is x[img][2 oy0 - 3 + ry][2 ox0 - 3 + cx][ci] (zero outside)
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    // Value e of a staged row sits 3 ix0 + e floats into its input row (NHWC rows are contiguous): no division per address,
    // in-row test = 0 <= 3 ix0 + e < 3 W.  ALL of a thread's 36 loads are issued before the first split, unconditionally
    // (clamped offsets, zeroed afterwards): left as one loop with the test around the load hipcc waits for every value
    // before the next load -- 28 dependent round trips per block, 215 us per launch at 16 x 512^2.
    bool sat = false;
    float v[C7_ROWS][4];
#pragma unroll
    for (int ry = 0; ry < C7_ROWS; ++ry) {
        const int iy = iy0 + ry;
        const float* xr = x + ((int64_t)img * H + ((iy >= 0 && iy < H) ? iy : 0)) * W * 3;  // (wave-uniform)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int off = ix0 * 3 + tid + 512 * m;
            v[ry][m] = xr[(off >= 0 && off < 3 * W) ? off : 0];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ry = 0; ry < C7_ROWS; ++ry) {
        const int iy = iy0 + ry;
        const bool rok = iy >= 0 && iy < H;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = tid + 512 * m, off = ix0 * 3 + e;
            const float val = (rok && e < 1551 && off >= 0 && off < 3 * W) ? v[ry][m] : 0.f;
            half_t h, l;
            sat |= fn_split(val, in_scale, h, l);
            if (e < C7_PROW) {
                xh[ry * C7_PROW + e] = h;
                xl[ry * C7_PROW + e] = l;
            }
        }
    }
    fn_flag_range(range_flag, sat);
    __syncthreads();
    // ---- products: wave = output row oy0 + (wave >> 2), columns ox0 + 64 (wave & 3) + 32 i + l31
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wr = wave >> 2, wc = (wave & 3) * 64;
    const char* wrow = wh + l31 * C7_WROW + hi * 16;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            half8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wr * 2 + ky) * C7_PROW + 6 * (wc + i * 32 + l31) + h2 * 16 + hi * 8;  // (even: 4-byte aligned)
                const uint32_t* ph = reinterpret_cast<const uint32_t*>(xh + off);
                const uint32_t* pl = reinterpret_cast<const uint32_t*>(xl + off);
                uintx4 vh = {ph[0], ph[1], ph[2], ph[3]}, vl = {pl[0], pl[1], pl[2], pl[3]};
                ah[i] = __builtin_bit_cast(half8_t, vh);
                al[i] = __builtin_bit_cast(half8_t, vl);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const half8_t*>(wrow + j * 32 * C7_WROW + (ky * 32 + h2 * 16) * 2);
                bl[j] = *reinterpret_cast<const half8_t*>(wrow + 64 * C7_WROW + j * 32 * C7_WROW + (ky * 32 + h2 * 16) * 2);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    // ---- epilogue: lane (l31, hi) holds channel 32 j + l31 of pixels 32 i + (r & 3) + 8 (r >> 2) + 4 hi of the wave's 64
    const int oy = oy0 + wr;
    const bool yok = oy < OH;
    int64_t orow[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = ox0 + wc + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            orow[i][r] = (yok && ox < OW) ? (((int64_t)img * OH + oy) * OW + ox) * 64 : -1;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r] * acc_scale;
                if (orow[i][r] >= 0) out[orow[i][r] + j * 32 + l31] = v;
                const double d = orow[i][r] >= 0 ? (double)v : 0.0;
                s1 += d;
                s2 += d * d;
            }
        if (stats) {  // (launcher: OW % 64 == 0 -> the wave's 64 pixels are rows [64 s, 64 s + 64) of the (n OH OW, 64) matrix)
            const double t1 = __shfl_xor(s1, 32, 64), t2 = __shfl_xor(s2, 32, 64);
            if (hi == 0 && yok && ox0 + wc < OW) {
                const int64_t slab = (((int64_t)img * OH + oy) * OW + ox0 + wc) >> 6;
                double* o = stats + (slab * 64 + j * 32 + l31) * 2;
                o[0] = s1 + t1;
                o[1] = s2 + t2;
            }
        }
    }
}

// Convex upsampling (gmflow.py:75-90): every fine pixel (8 y + ky, 8 x + kx) is a softmax-weighted mix of the 3 x 3 coarse
// neighbourhood of 8 * flow (zero padding, F.unfold order n = 3 dy + dx).  logits (B, h, w, 9 * 64) NHWC rows of the mask head
// (channel = n * 64 + ky * 8 + kx), flow (B, h w, 2) tokens, out (B, 2, 8 h, 8 w).  One thread per fine pixel.
__global__ __launch_bounds__(256) void fn_convex_upsample_kernel(const float* __restrict__ logits, const float* __restrict__ flow,
                                                                 float* __restrict__ out, int B, int h, int w) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * h * w * 64;
    if (idx >= total) return;
    const int sub = (int)(idx & 63), kx = sub & 7, ky = sub >> 3;
    const int64_t pix = idx >> 6;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((int64_t)w * h));
    const float* lg = logits + pix * 576 + sub;
    float e[9], mx = -3.0e38f;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        e[n] = lg[n * 64];
        mx = fmaxf(mx, e[n]);
    }
    float den = 0.f;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        e[n] = expf(e[n] - mx);
        den += e[n];
    }
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        const int yy = y + n / 3 - 1, xx = x + n % 3 - 1;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const float* f = flow + (((int64_t)b * h + yy) * w + xx) * 2;
            const float wgt = e[n] / den;
            ax = fmaf(wgt, 8.f * f[0], ax);
            ay = fmaf(wgt, 8.f * f[1], ay);
        }
    }
    const int64_t H8 = 8 * (int64_t)h, W8 = 8 * (int64_t)w;
    const int64_t o = ((int64_t)b * 2 * H8 + (8 * y + ky)) * W8 + 8 * x + kx;
    out[o] = ax;
    out[o + H8 * W8] = ay;
}

}  // namespace fresco

using namespace fresco;

extern "C" int fresco_fn_gemm(const void* a_hi, const void* a_lo, int64_t lda, const void* w_hi, const void* w_lo,
                              const float* bias, float* out, void* out_hi, void* out_lo, int64_t ldc, int64_t ldo, int M,
                              int N, int K, int act, float acc_scale, float split_scale, int n_img, int H, int W, int kh,
                              int kw, int stride, int pad, void* stats, const void* zeros, const int32_t* a_rows,
                              const int32_t* out_rows, int32_t* range_flag, int out_col_block, int64_t out_block_stride,
                              void* stream) {
    if (!zeros) return FRESCO_EINVAL;
    if ((a_rows || out_rows) && (kh > 0 || stats)) return FRESCO_EUNSUPPORTED;
    if (!a_hi || !a_lo || !w_hi || !w_lo || (!out && !out_hi) || (out_hi && !out_lo) || M <= 0 || N <= 0 || K <= 0)
        return FRESCO_EINVAL;
    if (K % 32 != 0 || lda % 8 != 0 || act < 0 || act > 2) return FRESCO_EUNSUPPORTED;
    if (out_col_block < 0 || (out_col_block > 0 && (!out || N % out_col_block != 0 || ldc < out_col_block || out_block_stride <= 0)))
        return FRESCO_EINVAL;
    if ((out && out_col_block == 0 && ldc < N) || (out_hi && ldo < N)) return FRESCO_EINVAL;
    if (out_hi && (N % 8 != 0 || ldo % 8 != 0)) return FRESCO_EUNSUPPORTED;
    FnConv cv = {0, 0, 1, 0, 0, 0, 0, 0, 0, 0};
    if (kh > 0) {
        if (kw <= 0 || stride <= 0 || pad < 0 || n_img <= 0 || H <= 0 || W <= 0) return FRESCO_EINVAL;
        const int cin = K / (kh * kw);
        if (cin * kh * kw != K || cin % 32 != 0 || lda < cin) return FRESCO_EUNSUPPORTED;
        const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
        if ((int64_t)n_img * OH * OW != M) return FRESCO_EINVAL;
        cv = FnConv{kh, kw, stride, pad, H, W, OH, OW, cin, (OH % 16 == 0 && OW % 16 == 0) ? 1 : 0};
        // fused InstanceNorm partial sums: a row block must not straddle two images
        if (stats && ((int64_t)OH * OW) % FN_BM != 0) return FRESCO_EUNSUPPORTED;
    } else if (stats) {
        return FRESCO_EUNSUPPORTED;
    } else if (lda < K) {
        return FRESCO_EINVAL;
    }
    hipStream_t st = as_stream(stream);
    ProfScope ps(FRESCO_PROF_FN_GEMM, M, N, K, kh, st);
    const half_t* ah = static_cast<const half_t*>(a_hi);
    const half_t* al = static_cast<const half_t*>(a_lo);
    const half_t* wh = static_cast<const half_t*>(w_hi);
    const half_t* wl = static_cast<const half_t*>(w_lo);
    half_t* oh = static_cast<half_t*>(out_hi);
    half_t* ol = static_cast<half_t*>(out_lo);
    double* sp = static_cast<double*>(stats);
    const int rb = (M + FN_BM - 1) / FN_BM;
    // 3 x 3 / stride 1 / pad 1 on whole 16 x 16 patches: the window-in-LDS form (FRESCO_FN_CONV_PATCH=0: the im2col form)
    static const bool patch_on = [] {
        const char* e = getenv("FRESCO_FN_CONV_PATCH");
        return !(e && e[0] == '0');
    }();
    static const bool xmap_on = [] {  // FRESCO_FN_XCD_MAP=0: column block outermost (the plain workgroup order)
        const char* e = getenv("FRESCO_FN_XCD_MAP");
        return !(e && e[0] == '0');
    }();
    const bool patch = patch_on && cv.tiled && kh == 3 && kw == 3 && stride == 1 && pad == 1;
#define FN_LAUNCH(BN_, PATCH_)                                                                                                \
    do {                                                                                                                      \
        const int lds = (PATCH_) ? 2 * 2 * 24 * 1024 + FN_NS * 2 * (BN_) * 64 : FN_NS * (2 * FN_BM * 64 + 2 * (BN_) * 64);    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fn_gemm_kernel<BN_, PATCH_>),                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                           \
        const int nb = (N + (BN_) - 1) / (BN_);                                                                               \
        hipLaunchKernelGGL((fn_gemm_kernel<BN_, PATCH_>), dim3(rb * nb), dim3(512), lds, st, ah, al, lda, cv, wh, wl, bias,   \
                           out, oh, ol, ldc, ldo, M, N, K, act, acc_scale, split_scale, sp, zeros, a_rows, out_rows,          \
                           range_flag, out_col_block, out_block_stride, nb | (xmap_on ? 1 << 16 : 0));                        \
    } while (0)
    if (N <= 64) {
        if (patch)
            FN_LAUNCH(64, true);
        else
            FN_LAUNCH(64, false);
    } else {
        if (patch)
            FN_LAUNCH(128, true);
        else
            FN_LAUNCH(128, false);
    }
#undef FN_LAUNCH
    return check_launch();
}

extern "C" size_t fresco_fn_colstats_workspace_bytes(int n_img, int rows, int C) {
    if (n_img <= 0 || rows <= 0 || C <= 0) return 0;
    const int slabs = (rows + 511) / 512;
    return (size_t)n_img * slabs * C * 2 * sizeof(double);
}

extern "C" int fresco_fn_colstats(const float* x, float* mean, float* rstd, void* workspace, size_t workspace_bytes,
                                  int n_img, int rows, int C, float eps, void* stream) {
    if (!x || !mean || !rstd || !workspace || n_img <= 0 || rows <= 0 || C <= 0) return FRESCO_EINVAL;
    if (C > 256 || n_img > 65535) return FRESCO_EUNSUPPORTED;
    if (workspace_bytes < fresco_fn_colstats_workspace_bytes(n_img, rows, C)) return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const int slabs = (rows + 511) / 512;
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(fn_colstats_partial_kernel, dim3(slabs, n_img), dim3(256), 0, st, x, part, rows, C, 512);
    hipLaunchKernelGGL(fn_colstats_final_kernel, dim3((C + 15) / 16, n_img), dim3(256), 0, st, part, mean, rstd, slabs, C, rows,
                       eps);
    return check_launch();
}

/* Finish InstanceNorm statistics from the partial sums fresco_fn_gemm left in `stats` (its `stats` argument): slabs =
 * rows / 64 per image (one per wave row of a 256-row block). */
extern "C" int fresco_fn_colstats_finish(const void* stats, float* mean, float* rstd, int n_img, int rows, int C, float eps,
                                         void* stream) {
    if (!stats || !mean || !rstd || n_img <= 0 || rows <= 0 || C <= 0 || rows % 256 != 0) return FRESCO_EINVAL;
    if (n_img > 65535) return FRESCO_EUNSUPPORTED;
    hipLaunchKernelGGL(fn_colstats_final_kernel, dim3((C + 15) / 16, n_img), dim3(256), 0, as_stream(stream),
                       static_cast<const double*>(stats), mean, rstd, rows / 64, C, rows, eps);
    return check_launch();
}

extern "C" int fresco_fn_prep(const float* x, const float* mean, const float* rstd, const float* residual, float* y,
                              void* out_hi, void* out_lo, int64_t M, int C, int ldo, int rows_per_img, int relu_a,
                              int relu_b, float split_scale, int32_t* range_flag, void* stream) {
    if (!x || M <= 0 || C <= 0 || (!y && !out_hi) || (out_hi && !out_lo) || ((mean != nullptr) != (rstd != nullptr)))
        return FRESCO_EINVAL;
    if (C % 4 != 0) return FRESCO_EUNSUPPORTED;
    if (!out_hi) ldo = C;
    if (ldo < C || ldo % 4 != 0 || (mean && rows_per_img <= 0)) return FRESCO_EINVAL;
    const int64_t n = M * (ldo / 4);
    hipLaunchKernelGGL(fn_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, mean, rstd,
                       residual, y, static_cast<half_t*>(out_hi), static_cast<half_t*>(out_lo), M, C, ldo,
                       rows_per_img > 0 ? rows_per_img : 1, relu_a, relu_b, split_scale, range_flag);
    return check_launch();
}

extern "C" int fresco_fn_layernorm(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                                   void* out_hi, void* out_lo, int64_t ldy, int64_t ldo, int64_t M, int C, float eps,
                                   float split_scale, int32_t* range_flag, void* stream) {
    if (!x || !gamma || !beta || M <= 0 || (!y && !out_hi) || (out_hi && !out_lo)) return FRESCO_EINVAL;
    if (C != 128) return FRESCO_EUNSUPPORTED;
    if ((y && ldy < C) || (out_hi && ldo < C)) return FRESCO_EINVAL;
    hipLaunchKernelGGL(fn_layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream), x, gamma, beta,
                       residual, y, static_cast<half_t*>(out_hi), static_cast<half_t*>(out_lo), ldy, ldo, M, eps, split_scale, range_flag);
    return check_launch();
}

extern "C" int fresco_fn_conv7_rgb(const float* x, const void* w_hi, const void* w_lo, float* out, void* stats, int n_img,
                                   int H, int W, int32_t* range_flag, void* stream) {
    if (!x || !w_hi || !w_lo || !out || n_img <= 0 || H <= 0 || W <= 0) return FRESCO_EINVAL;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (n_img > 65535 || (OH + 1) / 2 > 65535) return FRESCO_EUNSUPPORTED;
    // fused InstanceNorm partial sums: a wave's 64 pixels must be one 64-row slab of the (n OH OW, 64) matrix, and
    // fresco_fn_colstats_finish combines whole 256-row groups
    if (stats && (OW % 64 != 0 || ((int64_t)OH * OW) % 256 != 0)) return FRESCO_EUNSUPPORTED;
    const int lds = 2 * C7_ROWS * C7_PROW * 2 + 2 * 64 * C7_WROW;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fn_conv7_rgb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipStream_t st = as_stream(stream);
    ProfScope ps(FRESCO_PROF_FN_GEMM, n_img * OH * OW, 64, 147, 7, st);
    hipLaunchKernelGGL(fn_conv7_rgb_kernel, dim3((OW + 255) / 256, (OH + 1) / 2, n_img), dim3(512), lds, st, x,
                       static_cast<const half_t*>(w_hi), static_cast<const half_t*>(w_lo), out, static_cast<double*>(stats), H, W,
                       OH, OW, 64.f, 1.f / (64.f * 1024.f), range_flag);
    return check_launch();
}

extern "C" int fresco_fn_convex_upsample(const float* logits, const float* flow, float* out, int B, int h, int w, void* stream) {
    if (!logits || !flow || !out || B <= 0 || h <= 0 || w <= 0) return FRESCO_EINVAL;
    const int64_t total = (int64_t)B * h * w * 64;
    if ((total + 255) / 256 > 0x7fffffff) return FRESCO_EUNSUPPORTED;
    hipLaunchKernelGGL(fn_convex_upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), logits,
                       flow, out, B, h, w);
    return check_launch();
}
