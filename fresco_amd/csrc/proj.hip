// Fused linear projections of the attention processor (reference: attn.to_q / to_k / to_v / to_out[0],
// src/diffusion_hacked.py:201, 214-215, 260-261, 375):   out_j = x W_j^T (+ b_j),  j < nw <= 3.
//
// The reference issues one GEMM per projection; each re-reads x and, for M = B*HW rows against 320 or
// 640 features, sits far below both rooflines in the library GEMM (37 us per 65536 x 320 x 320 product =
// 2.3 TB/s, 360 TFLOP/s).  Here x is read ONCE: a wave keeps its 32 rows of x resident in registers as
// MFMA B fragments (K/16 fragments of 8 halfs) and streams slabs of 64 weight rows x 160 k through LDS -- the
// same swapped orientation as the flash kernel (out^T = W_tile x^T: one lane owns one row of x and ends up
// with 4 x 8 consecutive output features of a 64-feature tile).  Weight slabs are DMA'd global -> LDS (global_load_lds_dwordx4, issued as inline asm so that
// the compiler does not drain vmcnt in front of the fragment reads) into a 3-slot ring, TWO steps ahead, behind
// a counted s_waitcnt and one barrier per step; rows padded to an odd multiple of 16 B (conflict-free
// ds_read_b128); the weights (<= 2.4 MB) stay L2-resident.
//
// grid (ceil(M/256), splits): 8 waves x 32 rows; blockIdx.y takes a contiguous range of 64-feature tiles
// (splits > 1 only when M/256 alone cannot fill the chip).  What bounds the kernel is the rate at which a CU
// can stream the weight slabs into LDS (~25 GB/s per CU for LDS-DMA): every workgroup reads all of W, so rows per
// workgroup, not MFMA or HBM time, set the floor -- 256 rows (one workgroup per CU) halve that traffic vs 128.
// Measured (MI355X, M = 65536, K = 320, q,k,v in one launch): 56.6 us (round 1: 63.6); tools/proj_abl.hip: x loads +
// loop 11 us, + LDS fragment reads 10, + weight DMA 6.5, + MFMA 17, + output stores 12 -- the parts ADD UP: with one
// barrier-coupled workgroup per CU nothing overlaps.  Splitting the features over more workgroups (staggered
// lifetimes) re-reads x and is slower (2 / 3 / 5 splits: 68 / 78 / 94 us).  Output tiles are transposed through LDS
// so that a store instruction writes whole 128-byte lines (16-byte pieces of 32 rows each: +4 us).
// Round 4: a TILED form was written and measured against this one (both operands by swizzled LDS-DMA from where they
// live, 256 x 128 workgroup tiles of the concatenated output list, wave tiles 64 x 64, two workgroups per CU, 3-slot
// ring -- the structure of opt_fast.hip's Gram kernel; parity-green; commit history): 61.0 vs 52.7 us for q,k,v at
// (65536, 320), 56.5 vs 52.9 at (16384, 640), ties (+-1 us) on the gathered K|V launches and on to_out.  With K = 320
// a tile's K loop is 10 chunks: DMA prologue and store epilogue per tile outweigh what the deeper ring and the second
// workgroup per CU buy, and x is re-streamed 7.5 times.  Removed again; the resident-x form stays.
// Round 6: the resident x fragments are no longer loaded lane-per-row straight from global memory (64 rows one row stride
// apart per instruction: the texture-address path serves that at ~9 B/clk/CU -- profiles/r06_kvproj_ablation.txt) but in
// coalesced form (consecutive lanes = consecutive 16 bytes of a row) through a per-wave LDS region, all K chunks' loads in
// flight before the first goes through LDS, the weight slabs' DMA prologue requested in front of them.  Same fragments, same
// MFMA order: bit-identical outputs.  (16384, 640): single projection 29.9 -> 25.2 us, q,k,v 54.6 -> 49.8; (65536, 320),
// HBM-bound in that phase: 59.0 -> 58.8 (profiles/r06_ab_proj_x_staging.txt; -DFRESCO_PROJ_X_DIRECT=1 builds the old form).
#include "common.h"

namespace fresco {

template <int K, int NWV>
struct ProjCfg {
    static constexpr int KC = 160;              // K chunk staged per step (halfs)
    static constexpr int NKC = K / KC;          // steps per feature tile
    static constexpr int KS = KC / 16;          // MFMA k-steps per step
    static constexpr int NXF = K / 16;          // resident x fragments
    static constexpr int TF = 64;               // weight rows (output features) per tile
    static constexpr int ROWB = KC * 2 + 16;    // LDS bytes per weight row (21 chunks of 16 B: odd)
    static constexpr int CPR = ROWB / 16;
    static constexpr int NP = (TF * CPR + 63) / 64;  // 1 KiB DMA pieces per slab (the last one is partly pad)
    static constexpr int NPW_LO = NP / NWV, NPW_HI = (NP + NWV - 1) / NWV, NREM = NP % NWV;
    static constexpr int SLOT = NPW_HI * NWV * 1024;  // LDS bytes per ring slot
    static constexpr int NBUF = 3;  // ring slots: slabs are staged NBUF - 1 steps ahead
    static constexpr int RING_BYTES = NBUF * SLOT;
    // epilogue: a wave's 32 x 64 output tile is transposed through LDS so that every store instruction writes whole
    // 128-byte lines (8 lanes per row); rows padded to 144 B (conflict-free ds_write_b128 of 16 rows)
    static constexpr int OROW = TF * 2 + 16;
    static constexpr int OSCR = 32 * OROW;                // per wave
    // round 6: a wave's 32 rows of x reach their MFMA fragments through a per-wave LDS region -- coalesced global loads
    // (consecutive lanes = consecutive 16-byte pieces of a row), K chunk by K chunk, rows of XROW bytes (an odd number of
    // 16-byte chunks: conflict-free ds_read_b128).  The region is the wave's epilogue scratch later.
    static constexpr int XROW = KC * 2 + 16;
    static constexpr int XST = 32 * XROW;                 // per wave (>= OSCR)
    static constexpr int XPL = 32 * (KC / 8) / 64;        // staging loads per lane and K chunk
    static_assert(XST >= OSCR && (32 * (KC / 8)) % 64 == 0, "staging region");
    static constexpr int BIAS_OFF = RING_BYTES + NWV * XST;  // [nw][N] halfs follow
};

template <int N_>
__device__ __forceinline__ void proj_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N_) : "memory");
}

template <int K, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void linear_kernel(
    const half_t* __restrict__ x, int64_t x_ld, const int32_t* __restrict__ x_rows, const half_t* __restrict__ W0,
    const half_t* __restrict__ W1, const half_t* __restrict__ W2, const half_t* __restrict__ b0, const half_t* __restrict__ b1,
    const half_t* __restrict__ b2, half_t* __restrict__ out0, half_t* __restrict__ out1, half_t* __restrict__ out2,
    int64_t ld0, int64_t ld1, int64_t ld2, int M, int N, int nF, int tiles_per_split) {
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
    // wave takes k = hi*KC/2 + 8*ks .. +7 -- 160 contiguous bytes per lane and chunk instead of 16 out of
    // every 32.
    half8_t xf[Cfg::NXF];
    // A-tile row (lane & 31) is fed with weight row swap_bits_2_3(lane & 31): the 16 accumulator registers
    // of a lane then cover features 8*hi + (0..7) and 16 + 8*hi + (0..7) of its 32-row half of the tile
    const int frow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

    // DMA: per-lane byte offset of its chunk inside a (feature tile, K chunk) slab of W, computed once; pad
    // chunks (and the chunks past the slab in the last piece) re-read chunk 0
    uint32_t dma_off[Cfg::NPW_HI];
#pragma unroll
    for (int i = 0; i < Cfg::NPW_HI; ++i) {
        const int c = (i * NWV + wave_s) * 64 + lane;
        const int r = c / Cfg::CPR, dc = c % Cfg::CPR;
        dma_off[i] = (r < Cfg::TF && dc < Cfg::CPR - 1) ? (uint32_t)(r * K * 2 + dc * 16) : 0u;
    }
    const int many = wave_s < Cfg::NREM || Cfg::NREM == 0 ? 1 : 0;  // this wave issues NPW_HI pieces per slab
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
    // feature tile ft belongs to projection ft / tiles_per_out: every projection keeps its own (live) weight
    // matrix, nothing is stacked or cached on the host side.  The slab address is scalar arithmetic.
    const int tiles_per_out = N / Cfg::TF;
    const int64_t d1 = reinterpret_cast<const char*>(W1) - reinterpret_cast<const char*>(W0);
    const int64_t d2 = reinterpret_cast<const char*>(W2) - reinterpret_cast<const char*>(W0);
    // step s = (ft - ft0) * NKC + kc of this split's nsteps
    const int nsteps = (ft1 - ft0) * Cfg::NKC;
    auto stage = [&](int s, int slot) __attribute__((always_inline)) {
        const int ft = __builtin_amdgcn_readfirstlane(ft0 + s / Cfg::NKC), kc = s % Cfg::NKC;
        const int jw = ft / tiles_per_out;
        // (arithmetic instead of a three-way select: the compiler turns the select into a scratch-memory table)
        const char* src = reinterpret_cast<const char*>(W0) + (int64_t)(jw == 1) * d1 + (int64_t)(jw == 2) * d2 +
                          ((int64_t)(ft - jw * tiles_per_out) * Cfg::TF * K + kc * Cfg::KC) * 2;
        const uint32_t dstb = lds0 + slot * Cfg::SLOT + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < Cfg::NPW_HI; ++i) {
            if (i < Cfg::NPW_LO || many) {
                const uint32_t m0v = dstb + i * NWV * 1024;
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_off[i]), "s"(src), "s"(m0v)
                             : "memory");
            }
        }
    };
    // the slab of step s+1 has landed for everyone (own pieces: counted vmcnt, at most the newest slab's still in
    // flight -- output stores only make the wait stricter), and every wave is done with the slab of step s-1
    auto wait_barrier = [&](int keep) __attribute__((always_inline)) {  // keep = newer slabs that may stay in flight
        if (keep == 0)
            proj_wait_barrier<0>();
        else if (keep == 1 && many)
            proj_wait_barrier<Cfg::NPW_HI>();
        else if (keep == 1)
            proj_wait_barrier<Cfg::NPW_LO>();
        else if (many)
            proj_wait_barrier<2 * Cfg::NPW_HI>();
        else
            proj_wait_barrier<2 * Cfg::NPW_LO>();
    };
    constexpr int AHEAD = Cfg::NBUF - 1;

    // output cursor of this split: which tensor, which column
    int j = ft0 / tiles_per_out;
    int col = (ft0 % tiles_per_out) * Cfg::TF;

    stage(0, 0);
    if (AHEAD > 1 && nsteps > 1) stage(1, 1);
    if (AHEAD > 2 && nsteps > 2) stage(2, 2);
    // ---- x fragments (after the first weight slabs have been requested: their DMA runs under the x loads)
#ifndef FRESCO_PROJ_X_DIRECT
#define FRESCO_PROJ_X_DIRECT 0
#endif
    if (FRESCO_PROJ_X_DIRECT) {
        // (rounds 1-5, kept as an A/B build switch: every lane loads its own row's fragments straight from global memory --
        // 64 rows one row stride apart per instruction; the texture-address path serves that at ~9 B/clk/CU: 7.6 us for the
        // 164 KB of a workgroup at K = 320, 15 us for the 328 KB at K = 640)
        const int rr = row < M ? row : M - 1;
        const half_t* xp = x + (int64_t)(x_rows ? x_rows[rr] : rr) * x_ld + hi * (Cfg::KC / 2);
#pragma unroll
        for (int kc = 0; kc < Cfg::NKC; ++kc)
#pragma unroll
            for (int ks = 0; ks < Cfg::KS; ++ks)
                xf[kc * Cfg::KS + ks] = *reinterpret_cast<const half8_t*>(xp + kc * Cfg::KC + ks * 8);
    } else {
        // (x_rows: problem row m reads input row x_rows[m] -- the gathered form, e.g. K / V of the selected tokens only)
        // Piece q = j * 64 + lane of the wave's 32 x (KC / 8) pieces of a K chunk: row q / (KC / 8), piece q % (KC / 8) --
        // consecutive lanes read consecutive 16 bytes of a row (KC * 2 contiguous bytes per row: ~10 lines per instruction
        // instead of 64).  The next chunk's loads are in flight while this chunk goes through LDS.
        constexpr int PPR = Cfg::KC / 8;
        char* xst = smem + Cfg::RING_BYTES + wave * Cfg::XST;
        const int row0w = blockIdx.x * (NWV * 32) + wave * 32;
        uint32_t src16[Cfg::XPL];  // source piece index (16-byte units from x: x_ld % 8 == 0; tensors below 64 GB)
        int dst[Cfg::XPL];
#pragma unroll
        for (int j = 0; j < Cfg::XPL; ++j) {
            const int q = j * 64 + lane;
            const int r = q / PPR, dc = q % PPR;
            const int rr = min(row0w + r, M - 1);
            src16[j] = (uint32_t)((int64_t)(x_rows ? x_rows[rr] : rr) * (x_ld >> 3) + dc);
            dst[j] = r * Cfg::XROW + dc * 16;
        }
        // (plain macros, not lambdas over array references: hipcc keeps such arrays in scratch memory)
#define PROJ_LOAD_CHUNK(KCI, T)                                                              \
    _Pragma("unroll") for (int j = 0; j < Cfg::XPL; ++j)                                      \
        T[j] = *reinterpret_cast<const xu4_t*>(x + ((int64_t)src16[j] << 3) + (KCI) * Cfg::KC);
#define PROJ_FLUSH_CHUNK(KCI, T)                                                             \
    {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < Cfg::XPL; ++j)                                  \
            *reinterpret_cast<xu4_t*>(xst + dst[j]) = T[j];                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                               \
        __builtin_amdgcn_wave_barrier();                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                               \
        _Pragma("unroll") for (int ks = 0; ks < Cfg::KS; ++ks)                                \
            xf[(KCI) * Cfg::KS + ks] = *reinterpret_cast<const half8_t*>(                     \
                xst + l31 * Cfg::XROW + (hi * (Cfg::KC / 2) + ks * 8) * 2);                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                               \
        __builtin_amdgcn_wave_barrier(); /* the region is rewritten by the next chunk */     \
    }
        static_assert(Cfg::NKC == 2 || Cfg::NKC == 4, "staging schedule written for 2 or 4 K chunks");
        typedef unsigned int xu4_t __attribute__((ext_vector_type(4)));  // (a native vector: arrays of HIP's uint4 struct
                                                                         // stay in scratch memory across the fences)
        // every K chunk's loads are in flight before the first one goes through LDS: a staged register is dead when its
        // fragment register becomes live, so the total stays at the K / 8 x 4 registers the fragments need anyway
        xu4_t ta[Cfg::XPL], tb[Cfg::XPL];
        PROJ_LOAD_CHUNK(0, ta)
        PROJ_LOAD_CHUNK(1, tb)
        if constexpr (Cfg::NKC == 4) {
            xu4_t tc[Cfg::XPL], td[Cfg::XPL];
            PROJ_LOAD_CHUNK(2, tc)
            PROJ_LOAD_CHUNK(3, td)
            __builtin_amdgcn_sched_barrier(0);  // (hipcc sinks loads to their uses: keep the batches in front)
            PROJ_FLUSH_CHUNK(0, ta)
            PROJ_FLUSH_CHUNK(1, tb)
            PROJ_FLUSH_CHUNK(2, tc)
            PROJ_FLUSH_CHUNK(3, td)
        } else {
            __builtin_amdgcn_sched_barrier(0);
            PROJ_FLUSH_CHUNK(0, ta)
            PROJ_FLUSH_CHUNK(1, tb)
        }
#undef PROJ_LOAD_CHUNK
#undef PROJ_FLUSH_CHUNK
    }
    // biases -> LDS once (a global load inside the tile loop would make the compiler drain vmcnt there, DMA included)
    half_t* bias_s = reinterpret_cast<half_t*>(smem + Cfg::BIAS_OFF);
    const bool has_bias = b0 || b1 || b2;
    if (has_bias) {
        const int nwN = (nF / tiles_per_out) * N;
        for (int i = tid; i < nwN; i += NWV * 64) {
            const int jb = i / N;
            const half_t* bp = jb == 0 ? b0 : (jb == 1 ? b1 : b2);
            bias_s[i] = bp ? bp[i - jb * N] : (half_t)0.f;
        }
    }
    char* scr = smem + Cfg::RING_BYTES + wave * Cfg::XST;  // (the wave's x staging region, free by now)
    wait_barrier(min(nsteps - 1, AHEAD - 1));
    int slot = 0, s = 0;
    for (int ft = ft0; ft < ft1; ++ft) {
        floatx16 acc[2][2];  // [32-row half of the tile][two accumulation chains (even / odd k-steps)]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][0][r] = acc[t][1][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < Cfg::NKC; ++kc, ++s) {
            const int slot2 = slot >= 1 ? slot - 1 : Cfg::NBUF - 1;  // the slot of step s-1 takes the slab of step s+AHEAD
            if (s + AHEAD < nsteps) stage(s + AHEAD, slot2);
            const char* wr = smem + slot * Cfg::SLOT + frow * Cfg::ROWB + hi * Cfg::KC;  // hi * (KC/2) halfs
            {   // weight fragments are read PF MFMAs ahead of their use (hipcc's own schedule is read -> wait(0) -> MFMA per
                // fragment; r03: 60.7 -> 55.6 us for the L3 q,k,v launch, profiles/r03_ab_variants.txt)
                constexpr int NFR = Cfg::KS * 2, PF = 4;
                half8_t fr[NFR];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    fr[i] = *reinterpret_cast<const half8_t*>(wr + (i & 1) * 32 * Cfg::ROWB + (i >> 1) * 16);
#pragma unroll
                for (int i = 0; i < NFR; ++i) {
                    if (i + PF < NFR)
                        fr[i + PF] = *reinterpret_cast<const half8_t*>(wr + ((i + PF) & 1) * 32 * Cfg::ROWB + ((i + PF) >> 1) * 16);
                    const int ks = i >> 1, t = i & 1;
                    acc[t][ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], xf[kc * Cfg::KS + ks], acc[t][ks & 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);  // PF LDS reads up front,
#pragma unroll
                for (int i = 0; i < NFR - PF; ++i) {                 // then one MFMA, one LDS read, ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);  // and the last PF MFMAs
            }
            if (s + 1 < nsteps) {
                wait_barrier(min(nsteps - 2 - s, AHEAD - 1));  // slabs newer than s+1 that exist
            }
            slot = slot == Cfg::NBUF - 1 ? 0 : slot + 1;
        }
        // epilogue of the feature tile: + bias, fp16, transposed through the wave's LDS scratch: lane (row, hi) holds
        // the 16-byte chunks 2*c4 + hi of its row; store instruction i then writes rows 8i .. 8i+7 as whole lines
        half_t* op = out0;
        int64_t ld = ld0;
        if (j == 1) { op = out1; ld = ld1; }
        if (j == 2) { op = out2; ld = ld2; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                half8_t w;
                float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (has_bias) {
                    const half8_t b8 = *reinterpret_cast<const half8_t*>(bias_s + j * N + col + t * 32 + half * 16 + hi * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = (float)b8[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (half_t)(acc[t][0][half * 8 + e] + acc[t][1][half * 8 + e] + bv[e]);
                *reinterpret_cast<half8_t*>(scr + l31 * Cfg::OROW + ((t * 2 + half) * 2 + hi) * 16) = w;
            }
        {
            const int row0 = blockIdx.x * (NWV * 32) + wave * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = i * 8 + (lane >> 3);
                const int ch = lane & 7;
                // (keeps the 64-bit row addresses of the four stores INSIDE the tile loop: hoisted, they cost the K = 640
                // instantiation -- 160 resident x registers -- a spill that is reloaded in every tile's store path)
                asm volatile("" : "+v"(r));
                const half8_t w = *reinterpret_cast<const half8_t*>(scr + r * Cfg::OROW + ch * 16);
                if (row0 + r < M) *reinterpret_cast<half8_t*>(op + (int64_t)(row0 + r) * ld + col + ch * 8) = w;
            }
        }
        col += Cfg::TF;
        if (col == N) {
            col = 0;
            ++j;
        }
    }
}

template <int K, int NWV>
static int launch_linear(const half_t* x, int64_t x_ld, const int32_t* x_rows, const half_t* const* W, const half_t* const* bias,
                         half_t* out0, half_t* out1, half_t* out2, int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M,
                         int N, hipStream_t st) {
    using Cfg = ProjCfg<K, NWV>;
    // (per device and cheap: set on every launch rather than cached in a process-global flag)
    const int lds_bytes = Cfg::BIAS_OFF + nw * N * 2;
    if (lds_bytes > 160 * 1024) return FRESCO_EUNSUPPORTED;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_kernel<K, NWV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int nF = nw * N / Cfg::TF;
    const int row_blocks = (M + NWV * 32 - 1) / (NWV * 32);
    // enough workgroups for two rounds of the 256 CUs; every extra split re-reads x once
    int splits = (2048 / NWV + row_blocks - 1) / row_blocks;
    if (splits > nF) splits = nF;
    if (splits < 1) splits = 1;
    const int tiles_per_split = (nF + splits - 1) / splits;
    splits = (nF + tiles_per_split - 1) / tiles_per_split;
    ProfScope ps(FRESCO_PROF_LINEAR, M, N, K, nw, st);
    hipLaunchKernelGGL((linear_kernel<K, NWV>), dim3(row_blocks, splits), dim3(NWV * 64), lds_bytes, st, x, x_ld, x_rows, W[0], W[1],
                       W[2], bias[0], bias[1], bias[2], out0, out1, out2, ld0, ld1, ld2, M, N, nF, tiles_per_split);
    return check_launch();
}

}  // namespace fresco

using namespace fresco;

static int linear_dispatch(const void* x, int64_t x_ld, const int32_t* x_rows, const void* W0, const void* W1,
                           const void* W2, const void* b0, const void* b1, const void* b2, void* out0, void* out1,
                           void* out2, int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M, int N, int K, void* stream) {
    if (!x || !W0 || !out0 || nw < 1 || nw > 3 || M <= 0 || N <= 0 || K <= 0) return FRESCO_EINVAL;
    if ((nw > 1 && (!out1 || !W1)) || (nw > 2 && (!out2 || !W2))) return FRESCO_EINVAL;
    if (x_ld < K || x_ld % 8 != 0) return FRESCO_EINVAL;
    if (ld0 < N || ld0 % 8 != 0 || (nw > 1 && (ld1 < N || ld1 % 8 != 0)) || (nw > 2 && (ld2 < N || ld2 % 8 != 0)))
        return FRESCO_EINVAL;
    if (N % 64 != 0 || (K != 320 && K != 640)) return FRESCO_EUNSUPPORTED;
    if ((int64_t)(M + 127) / 128 > 0x7fffffff) return FRESCO_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const half_t* xh = static_cast<const half_t*>(x);
    const half_t* wh[3] = {static_cast<const half_t*>(W0), static_cast<const half_t*>(W1), static_cast<const half_t*>(W2)};
    const half_t* bh[3] = {static_cast<const half_t*>(b0), static_cast<const half_t*>(b1), static_cast<const half_t*>(b2)};
    half_t* o0 = static_cast<half_t*>(out0);
    half_t* o1 = static_cast<half_t*>(out1);
    half_t* o2 = static_cast<half_t*>(out2);
    if (K == 320) return launch_linear<320, 8>(xh, x_ld, x_rows, wh, bh, o0, o1, o2, ld0, ld1, ld2, nw, M, N, st);
    return launch_linear<640, 8>(xh, x_ld, x_rows, wh, bh, o0, o1, o2, ld0, ld1, ld2, nw, M, N, st);
}

extern "C" int fresco_linear(const void* x, int64_t x_ld, const void* W0, const void* W1, const void* W2,
                             const void* b0, const void* b1, const void* b2, void* out0, void* out1, void* out2,
                             int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M, int N, int K, void* stream) {
    return linear_dispatch(x, x_ld, nullptr, W0, W1, W2, b0, b1, b2, out0, out1, out2, ld0, ld1, ld2, nw, M, N, K, stream);
}

extern "C" int fresco_linear_rows(const void* x, int64_t x_ld, const int32_t* x_rows, const void* W0, const void* W1,
                                  const void* W2, const void* b0, const void* b1, const void* b2, void* out0,
                                  void* out1, void* out2, int64_t ld0, int64_t ld1, int64_t ld2, int nw, int M, int N,
                                  int K, void* stream) {
    if (!x_rows) return FRESCO_EINVAL;
    return linear_dispatch(x, x_ld, x_rows, W0, W1, W2, b0, b1, b2, out0, out1, out2, ld0, ld1, ld2, nw, M, N, K, stream);
}
