// Dense fp16 attention with shared key groups for FRESCO's spatial-guided and efficient
// cross-frame passes (reference: src/diffusion_hacked.py:225-247, 250-254, 281-285, 303-305, 371).
//
// Two kernels:
//   kv_pack_kernel  : gathers the selected K / V rows of one key group and writes, per 64-key tile, the
//                     exact LDS image the MFMA loop consumes (K fragments ‖ V^T fragments, 16-byte chunks
//                     in ds_read_b128-conflict-free order).  HBM-bound, a few MB.
//   attn_flash_kernel: flash-style attention on v_mfma_f32_32x32x16_f16.  One wave owns 32*QB query rows, a
//                     workgroup 8 waves = two per SIMD, 256 registers each.  S^T = K Q^T is computed
//                     "swapped", so every lane holds the scores of ONE query (its column of the 32x32 C tile):
//                     the softmax needs no cross-lane traffic and the exponentiated scores, packed to fp16,
//                     already ARE the B operand of O^T = V^T P^T (the key order inside each 16-key MFMA step is
//                     the C-tile row order; kv_pack writes V^T in it).
//
// What bounds this kernel on gfx950 (tools/ubench_rates.hip, profiles/r02_attn_experiments.txt): not the matrix
// pipe alone but the SIMD's one VALU/issue port, which v_exp_f32 holds for 8.25 cycles, v_cvt_pk_f16_f32 for
// 4.3 and every MFMA issue for ~7 -- from either of the two waves of the SIMD; MFMA execution (32 cycles per
// 32x32x16) overlaps with the partner wave's VALU work, but two waves left to themselves run in lockstep
// (both in their MFMA block, then both in their softmax block: MFMA time + VALU time, no overlap).  Per 64 keys
// x 64 queries at D = 40: port = 64 exp + 32 cvt + 28 MFMA issues ~ 880 cycles, matrix pipe = 28 x 32 = 896.
// Hence: as few and as large MFMAs as possible (32x32x16 for both products; a 16x16x32 PV product would save
// pipe time but costs more issues plus a permlane per P register: measured slower), no per-score VALU besides
// exp and cvt (the scale is folded into Q, the running max rides in the QK product, the row sum in the PV
// product, see below), and an explicit PING-PONG of the two waves of a SIMD:
//   * waves 0-3 (group A) and 4-7 (group B) run the same loop body  [ring barrier | V^T reads, softmax(u) |
//     K reads, PV(u), QK(u+1)], but A passes the workgroup barrier AFTER its softmax and B BEFORE it: the
//     barrier therefore releases A's matrix block together with B's softmax block and vice versa, and keeps
//     them half a step apart for the whole key loop (s_setprio raises the matrix block);
//   * key packs (K of tile u+1 next to V^T of tile u: exactly what one loop body reads) arrive by DMA
//     (global_load_lds_dwordx4, linear 1 KiB copies) into a 4-slot LDS ring, three steps ahead, behind counted
//     vmcnt waits and that ONE barrier per step;
//   * K fragments are read under the PV MFMAs, V^T fragments under the softmax: no MFMA waits on LDS;
//   * two query blocks per wave at D <= 48: every fragment read feeds two MFMAs;
//   * the common case (no diagonal bias, scale folded, >= 3 tiles ahead) is an instantiation of its own without
//     the scalar branches of the rare passes (a taken branch costs an instruction-fetch bubble).
// The softmax bookkeeping rides in the MFMAs wherever the head dim leaves room: a ones ROW in V^T makes the
// PV product deliver the row sum, a ones COLUMN in K against -m in Q's spare column makes the QK product
// subtract the running max.  Per wave, from the key norms kv_pack records (Cauchy-Schwarz bound on the
// logits): the max search is dropped when no exponent can leave fp16 range, and the exponent scale is folded
// into the fp16 Q only while that costs no more than P's own rounding.
//
// MFMA 32x32x16 f16 operand layout (gfx950): lane l supplies 8 consecutive k for row/col (l & 31), k-chunk
// (l >> 5); C/D: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15.
#include "attn_cfg.h"
#include <type_traits>

namespace fresco {

// CUs of the current device (cached per device): the 256-row / 512-row workgroup choice below depends on it
static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}


// ---------------------------------------------------------------------------------------------
// pack: grid (nT, H, G), 256 threads.  Pack p of the image = K fragments of tile p || V^T fragments of tile p - 1: what
// ONE loop step of attn_flash_kernel reads (PV(u) next to QK(u+1)); nT + 1 packs.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void kv_pack_kernel(const half_t* __restrict__ k,
                                                       const half_t* __restrict__ v,
                                                       const int32_t* __restrict__ kv_rows,
                                                       char* __restrict__ img, float* __restrict__ ktmax,
                                                       int H, int M, int nT, int64_t group_rows,
                                                       int64_t kv_ld) {
    using Cfg = AttnCfg<D>;
    const int tile = blockIdx.x, h = blockIdx.y, g = blockIdx.z;
    __shared__ int32_t rows[64];
    __shared__ __attribute__((aligned(16))) half_t ks[64][D + 8];  // +8 halfs: 16-B aligned rows
    __shared__ __attribute__((aligned(16))) half_t vs[64][D + 8];

    if (threadIdx.x < 64) {
        const int m = tile * 64 + threadIdx.x;
        int32_t r = -1;
        if (m < M) r = kv_rows ? kv_rows[m] : m;
        rows[threadIdx.x] = r;
    }
    __syncthreads();

    // stage the 64 x D slabs of K and V (each row D halfs contiguous in global memory)
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (int c = threadIdx.x; c < 64 * (D / 8); c += 256) {
        const int row = c / (D / 8), dc = c % (D / 8);
        uint4 kv = zero, vv = zero;
        const int32_t r = rows[row];
        if (r >= 0) {
            const int64_t off = ((int64_t)g * group_rows + r) * kv_ld + h * D + dc * 8;
            kv = *reinterpret_cast<const uint4*>(k + off);
            vv = *reinterpret_cast<const uint4*>(v + off);
        }
        *reinterpret_cast<uint4*>(&ks[row][dc * 8]) = kv;
        *reinterpret_cast<uint4*>(&vs[row][dc * 8]) = vv;
    }
    __syncthreads();

    char* dst = img + ((int64_t)(g * H + h) * (nT + 1) + tile) * Cfg::TILE;
    // K chunks
    for (int c = threadIdx.x; c < Cfg::NKS * 128; c += 256) {
        const int key = c & 63, d0 = (c >> 6) * 8;
        uint4 val = zero;
        if (d0 < D)
            val = *reinterpret_cast<const uint4*>(&ks[key][d0]);
        else if (Cfg::MCOL && d0 == D && rows[key] >= 0)
            val.x = 0x3C00u;  // K[key][D] = 1.0: with Q[query][D] = -m the QK MFMA delivers  q.k - m
        *reinterpret_cast<uint4*>(dst + (int64_t)c * 16) = val;
    }
    // V^T chunks
    for (int c = threadIdx.x; c < 8 * Cfg::DPV; c += 256) {
        const int d = c % Cfg::DPV, cc = (c / Cfg::DPV) & 1, kc = c / (2 * Cfg::DPV);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = kc * 16 + (e & 3) + 8 * (e >> 2) + 4 * cc;
            half_t val = (half_t)0;
            if (d < D)
                val = vs[key][d];
            else if (Cfg::ONES && d == D && rows[key] >= 0)
                val = (half_t)1;  // ones row: only real keys count towards the softmax denominator
            o[e] = val;
        }
        *reinterpret_cast<half8_t*>(dst + Cfg::TILE + Cfg::KTILE + (int64_t)c * 16) = o;
    }
    // largest squared key norm of the tile (four threads per key, fixed summation order): the flash kernel
    // bounds every logit of a query by |q| max|k| (Cauchy-Schwarz) and drops the running-max search when
    // that bound cannot leave fp16 range
    {
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
        float n2 = 0.f;
        for (int dc = part; dc < D / 8; dc += 4) {
            const half8_t kk = *reinterpret_cast<const half8_t*>(&ks[row][dc * 8]);
#pragma unroll
            for (int e = 0; e < 8; ++e) n2 = fmaf((float)kk[e], (float)kk[e], n2);
        }
        n2 += __shfl_xor(n2, 1, 64);
        n2 += __shfl_xor(n2, 2, 64);
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) n2 = fmaxf(n2, __shfl_xor(n2, off, 64));
        __shared__ float wmax[4];
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = n2;
        __syncthreads();
        if (threadIdx.x == 0)
            ktmax[(int64_t)(g * H + h) * nT + tile] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    }
}

// ---------------------------------------------------------------------------------------------
// kvproj_pack (round 6): the K | V projection of the SELECTED rows and the pack in ONE launch, for layer calls whose only
// reader of K and V is the cross-frame pass (cross-frame-only steps: 7 of the 15 of the schedule).  Before: fresco_linear_rows
// (K | V of the 2 x M gathered hidden rows -> HBM, 15 - 18 us) then kv_pack_kernel (gather again, transpose, pad -> image,
// 9 - 13 us), two latency-bound launches in front of the flash kernel.  Here a workgroup owns one 64-key tile of one CFG
// half and 160 output features of K and of V (4 heads at D = 40, 2 at D = 80): the 64 gathered hidden rows are staged in
// LDS once (odd 16-byte row stride: conflict-free fragment reads), each of the 5 waves streams its 32 weight rows of W_k
// and of W_v straight from L2 into MFMA fragments (no LDS: nothing is shared between waves), and the accumulators ARE the
// image's pieces -- K as C[feature][key] (A = W_k, B = x: a lane's 4 consecutive features of a key are 8 bytes of the
// key's chunk), V^T as C[key][feature] (A = x, B = W_v: a lane's registers 8 kc' .. 8 kc' + 7 are the 8 keys of one
// 16-byte V^T chunk, in the C-tile key order the flash kernel's PV product consumes).  The constant parts of the image
// (ones column of K, ones row / zero rows of V^T) and max |k|^2 per (head, tile) are written alongside (partial sums per
// 4 features in LDS, added in a fixed order: run-to-run identical).  K and V never exist in HBM.
// Numerics: one fp32 accumulation chain per output in natural k order (linear_kernel: two chains, permuted k) -- the fp16
// K / V values can differ from the two-launch path's in the last place; parity is against the oracle, as everywhere.
// grid (nT, H*D/160, G), 320 threads.
// ---------------------------------------------------------------------------------------------
template <int KIN, int D>
struct KvProjCfg {
    static constexpr int ROWB = KIN * 2 + 16;  // LDS bytes per staged hidden row: an odd number of 16-byte chunks
    static constexpr int HPW = 160 / D;        // heads per workgroup
    static constexpr int NPART = D / 4;        // partial sums of |k|^2 per key and head (one per 4 features)
    static constexpr int XS_BYTES = 64 * ROWB;
    static constexpr int LDS_BYTES = XS_BYTES + HPW * NPART * 64 * 4 + 64 * 4;
};

template <int KIN, int D>
__global__ __launch_bounds__(320, KIN == 320 ? 3 : 2) void kvproj_pack_kernel(const half_t* __restrict__ x, int64_t x_ld,
                                                           const int32_t* __restrict__ x_rows,
                                                           const half_t* __restrict__ Wk,
                                                           const half_t* __restrict__ Wv, char* __restrict__ img,
                                                           float* __restrict__ ktmax, int H, int M, int nT) {
    using Cfg = AttnCfg<D>;
    using PC = KvProjCfg<KIN, D>;
    constexpr int ROWB = PC::ROWB, HPW = PC::HPW, NPART = PC::NPART;
    constexpr int NKS = KIN / 16;         // MFMA k-steps
    // k-steps per pipeline stage = one 128-byte line of every weight row: a lane (row, hi) takes the 64-byte half line
    // [64 hi, 64 hi + 64) of the line as its fragments of the stage's four k-steps (the contraction order is free as long as
    // the hidden-row fragments follow it) -- four back-to-back loads of one line, of which three hit in L1; with the natural
    // order (16 bytes of every other 32) a line is touched by four instructions one pipeline step apart and has left the
    // 32 KB L1 in between: weight loads 10 us of this launch's 24 (profiles/r06_kvproj_ablation.txt)
    constexpr int SK = 4;
    constexpr int NST = NKS / SK;
    constexpr int CPR = KIN / 8;          // 16-byte chunks per hidden row
    constexpr int NXL = 64 * CPR / 320;   // staging loads per thread
    static_assert(NKS % SK == 0 && (64 * CPR) % 320 == 0 && 160 % D == 0 && D % 8 == 0, "shapes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    float* n2p = reinterpret_cast<float*>(smem + PC::XS_BYTES);
    int32_t* rows = reinterpret_cast<int32_t*>(n2p + HPW * NPART * 64);
    const int tile = blockIdx.x, fg = blockIdx.y, g = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    if (tid < 64) {
        const int m = tile * 64 + tid;
        rows[tid] = m < M ? x_rows[(int64_t)g * M + m] : -1;
    }
    const int ft = fg * 5 + wave;  // this wave's 32-feature tile of K and of V
    const half_t* wkp = Wk + (int64_t)(ft * 32 + l31) * KIN + hi * 32;
    const half_t* wvp = Wv + (int64_t)(ft * 32 + l31) * KIN + hi * 32;
    // Software pipeline, one stage ahead: the weight fragments of stage s + 1 are requested before the products of stage s
    // (hipcc on its own sinks every load to its use: load -> wait -> MFMA, one L2 round trip per k-step -- 28 us for this
    // launch; the sched_barriers pin the batches).  Stage 0 is requested before the hidden rows are gathered.
    half8_t fk[2][SK], fv[2][SK];
    auto fetch = [&](int st, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SK; ++i) {
            fk[buf][i] = *reinterpret_cast<const half8_t*>(wkp + st * 64 + i * 8);
            fv[buf][i] = *reinterpret_cast<const half8_t*>(wvp + st * 64 + i * 8);
        }
    };
#ifndef KVP_ABL
#define KVP_ABL 0
#endif
    if (!(KVP_ABL & 2)) fetch(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // rows[]
    {   // gather the 64 hidden rows into LDS: all of a thread's loads in flight, then its writes
        uint4 xv[NXL];
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int c = tid + i * 320;
            const int row = c / CPR, dc = c % CPR;
            const int32_t r = rows[row];
            xv[i] = make_uint4(0, 0, 0, 0);
            if (r >= 0 && !(KVP_ABL & 4)) xv[i] = *reinterpret_cast<const uint4*>(x + (int64_t)r * x_ld + dc * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int c = tid + i * 320;
            *reinterpret_cast<uint4*>(xs + (c / CPR) * ROWB + (c % CPR) * 16) = xv[i];
        }
    }
    floatx16 ak[2], av[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[b][r] = av[b][r] = 0.f;
    __syncthreads();
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int buf = st & 1;
        if (st + 1 < NST && !(KVP_ABL & 2)) fetch(st + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SK; ++i) {
            const int kb = (st * 64 + hi * 32 + i * 8) * 2;  // byte offset inside a staged row (same k order as the weights)
            const half8_t x0 = *reinterpret_cast<const half8_t*>(xs + l31 * ROWB + kb);
            const half8_t x1 = *reinterpret_cast<const half8_t*>(xs + (32 + l31) * ROWB + kb);
            ak[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk[buf][i], x0, ak[0], 0, 0, 0);
            ak[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fk[buf][i], x1, ak[1], 0, 0, 0);
            av[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, fv[buf][i], av[0], 0, 0, 0);
            av[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, fv[buf][i], av[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((KVP_ABL & 1) && ak[0][0] + av[1][3] != 12345.f) return;  // (ablation: no epilogue)
    // ---- K pieces: lane (key l31 of block b, hi), registers 4j .. 4j+3 = features 32 ft + 8 j + 4 hi + (0..3)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int key = b * 32 + l31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f0 = ft * 32 + 8 * j + 4 * hi;
            const int head = f0 / D, dd = f0 - head * D;
            half4_t w;
            float n2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w[e] = (half_t)ak[b][4 * j + e];
                n2 = fmaf((float)w[e], (float)w[e], n2);
            }
            char* dst = img + ((int64_t)(g * H + head) * (nT + 1) + tile) * Cfg::TILE;
            *reinterpret_cast<half4_t*>(dst + ((dd >> 3) * 64 + key) * 16 + (dd & 7) * 2) = w;
            n2p[((head - fg * HPW) * NPART + (dd >> 2)) * 64 + key] = n2;
        }
    }
    // ---- V^T pieces: lane (feature 32 ft + l31, hi), registers 8 kc' .. 8 kc' + 7 = keys of chunk (kc = 2 b + kc', cc = hi)
    {
        const int f = ft * 32 + l31;
        const int head = f / D, d = f - head * D;
        char* dst = img + ((int64_t)(g * H + head) * (nT + 1) + tile + 1) * Cfg::TILE + Cfg::KTILE;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                half8_t w;
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (half_t)av[b][8 * kq + e];
                *reinterpret_cast<half8_t*>(dst + (((2 * b + kq) * 2 + hi) * Cfg::DPV + d) * 16) = w;
            }
    }
    // ---- constant parts of the image for this workgroup's heads
    {
        constexpr int KPAD = (Cfg::DPK - D) / 8;         // pad chunks of K per key (the first one carries the ones column)
        constexpr int VPAD = Cfg::DPV - D;               // pad rows of V^T (the first one is all ones)
        constexpr int PER_HEAD = KPAD * 64 + VPAD * 8;
        for (int i = tid; i < HPW * PER_HEAD; i += 320) {
            const int hl = i / PER_HEAD, c = i % PER_HEAD;
            const int head = fg * HPW + hl;
            char* base = img + ((int64_t)(g * H + head) * (nT + 1) + tile) * Cfg::TILE;
            if (c < KPAD * 64) {
                const int pc = c / 64, key = c % 64;
                uint4 val = make_uint4(0, 0, 0, 0);
                if (Cfg::MCOL && pc == 0 && rows[key] >= 0) val.x = 0x3C00u;  // K[key][D] = 1.0 (running max rides in the QK MFMA)
                *reinterpret_cast<uint4*>(base + ((D / 8 + pc) * 64 + key) * 16) = val;
            } else {
                const int c2 = c - KPAD * 64;
                const int pr = c2 / 8, kcc = c2 % 8;     // pad row, (kc, cc) chunk
                const int kc = kcc >> 1, cc = kcc & 1;
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = kc * 16 + (e & 3) + 8 * (e >> 2) + 4 * cc;
                    o[e] = (Cfg::ONES && pr == 0 && rows[key] >= 0) ? (half_t)1 : (half_t)0;
                }
                *reinterpret_cast<half8_t*>(base + Cfg::TILE + Cfg::KTILE + (kcc * Cfg::DPV + D + pr) * 16) = o;
            }
        }
    }
    __syncthreads();
    // ---- max |k|^2 of the tile per head: parts added in a fixed order, maximum over the 64 keys
    if (wave < HPW) {
        float n2 = 0.f;
#pragma unroll
        for (int p_ = 0; p_ < NPART; ++p_) n2 += n2p[(wave * NPART + p_) * 64 + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) n2 = fmaxf(n2, __shfl_xor(n2, off, 64));
        if (lane == 0) ktmax[(int64_t)(g * H + fg * HPW + wave) * nT + tile] = n2;
    }
}

// ---------------------------------------------------------------------------------------------
// flash attention: grid (H * nQblk * B), 512 threads = 8 waves x QB blocks of 32 query rows
// blockIdx.x = (b * nQblk + qblk) * H + h   -> head h lands on XCD (h % 8): each XCD's L2 holds
// only its own heads' packed key images.
//
// Two waves share every SIMD, and what they share is the matrix pipe and the VALU port: left alone, the two
// run their MFMA phases together (each at half rate) and then their softmax phases together, in lockstep
// (measured: MFMA time + softmax time, no overlap).  So the workgroup is two groups of four waves (one per
// SIMD each) that are held half a tile apart by WHERE in the tile they execute the workgroup's one barrier:
//     group A:  softmax(u) | barrier | PV(u)  QK(u+1)            -> its MFMA block follows the barrier
//     group B:  barrier | softmax(u)  PV(u)  QK(u+1)             -> its softmax follows the barrier
// between two barriers a SIMD sees [A: 28 MFMAs || B: exp/cvt] and then [A: exp/cvt || B: 28 MFMAs].  The
// softmax block (64 exp + 32 cvt, plus the issue slots the partner's MFMAs take) is a little shorter than the
// MFMA block (28 x 32 cycles), so the matrix pipe is the pacing resource of both halves.
//
// Key tiles: pack p of the image = K fragments of tile p ‖ V^T fragments of tile p-1, what step u = p-1 of the
// loop reads (PV(u), QK(u+1)).  Packs arrive by DMA into a 4-slot LDS ring, three steps ahead; a wave waits for
// its own pieces of the pack after next (counted vmcnt) before the barrier, so every fragment read finds its
// data landed one barrier earlier and no MFMA waits on global memory.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void ring_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int D, int QB>
__global__ __launch_bounds__(512, 2) void attn_flash_kernel(const half_t* __restrict__ q,
                                                          const char* __restrict__ img,
                                                          const float* __restrict__ ktmax,
                                                          half_t* __restrict__ out, int B, int H, int Lq,
                                                          int M, int nT, int batch_per_group,
                                                          float scale_log2, float diag_bias_log2, int64_t q_ld) {
    using Cfg = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 256 * QB;  // query rows per workgroup

    const int nQblk = (Lq + ROWS - 1) / ROWS;
    const unsigned blk = blockIdx.x;
    const int h = blk % H;
    const int qblk = (blk / H) % nQblk;
    const int b = blk / (H * nQblk);
    const int g = b / batch_per_group;
    const int C = H * D;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow0 = qblk * ROWS + wave * 32 * QB + l31;  // row of query block 0; block j: + 32*j
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int grpB = wave_s >= 4 ? 1 : 0;  // (flags are ints from scalar values: the branches on them stay scalar)

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
    //    exponent by at most 2^-12 * c|q||k|: at FOLD_MAX = 24 a WORST-CASE 0.6 % of P, ~12 x P's own fp16 rounding.
    //    The limit is therefore empirical, not derived: tools/fold_margin.py emulates the kernel's arithmetic and
    //    finds the folded form at 0.30 of the 1e-3 parity bar for N(0,1) keys and at the exact form's error for keys
    //    aligned to the query (attn_cfg.h); test_attention_fold_limit_structured covers non-Gaussian q, k (a few
    //    dominant channels, correlated q / k) at the limit.  Beyond FOLD_MAX Q stays exact and every score is
    //    multiplied by c in fp32.
    //  * no running-max search (nomax, below) when the bound cannot leave fp16 range.
    // Accumulator units u: exponent argument = cmul * u, with (qs, cmul) = (c, 1) folded or (1, c) exact.
    float kmax;
    {
        const float* km = ktmax + (int64_t)(g * H + h) * nT;
        float k2 = 0.f;
        for (int i = lane; i < nT; i += 64) k2 = fmaxf(k2, km[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) k2 = fmaxf(k2, __shfl_xor(k2, off, 64));
        kmax = sqrtf(k2);
    }
    bool fold_ok = true;
#pragma unroll
    for (int j = 0; j < QB; ++j) fold_ok = fold_ok && (scale_log2 * sqrtf(q2[j]) * kmax <= FOLD_MAX);
    // (flags are ints read from scalar values: the branches on them stay scalar branches)
    const int folded = __builtin_amdgcn_readfirstlane((int)__all(fold_ok));
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

    // ---- staging: global -> LDS by DMA (global_load_lds_dwordx4), no register round trip ---------
    // A pack IS its LDS image, so it is NP linear 1 KiB copies; wave w issues pieces w, w+8, ...
    // (destination = wave-uniform M0 base + lane*16, source = scalar base + one per-lane offset).  The DMA
    // is inline asm on purpose: the compiler must not see these LDS writes, or it would drain vmcnt to zero
    // in front of every fragment read; ordering is by the counted s_waitcnt + s_barrier of `ring_sync`.
    constexpr int NPW_LO = Cfg::NP / 8, NPW_HI = (Cfg::NP + 7) / 8, NREM = Cfg::NP % 8;
    const int many = wave_s < NREM ? 1 : 0;  // this wave issues NPW_HI pieces per pack (else NPW_LO)
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
    const char* src = img + (int64_t)(g * H + h) * (nT + 1) * Cfg::TILE;
    const uint32_t lane_off = (wave * 64 + lane) * 16;
    auto stage = [&](int p, int slot) __attribute__((always_inline)) {  // pack p -> ring slot
        const char* sp = src + (int64_t)p * Cfg::TILE;
        const uint32_t dstb = lds0 + slot * Cfg::TILE + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < NPW_HI; ++i) {
            if (i < NPW_LO || many) {
                const uint32_t m0v = dstb + i * 8192;
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sp), "s"(m0v)
                             : "memory");
                sp += 8192;
            }
        }
    };
    // wait until at most `keep` of this wave's newest tiles are still in flight, then the workgroup barrier
    auto wait_barrier = [&](int keep) __attribute__((always_inline)) {
        if (keep == 0) {
            ring_wait_barrier<0>();
        } else if (many) {
            if (keep == 1) ring_wait_barrier<NPW_HI>(); else ring_wait_barrier<2 * NPW_HI>();
        } else {
            if (keep == 1) ring_wait_barrier<NPW_LO>(); else ring_wait_barrier<2 * NPW_LO>();
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

    // per-lane fragment offset inside a ring slot (K: + (ks*128 + kb*32)*16; V^T: + KTILE + (kc*2*DPV + db*32)*16)
    const int koff = (hi * 64 + l31) * 16;
    const int voff = Cfg::KTILE + (hi * Cfg::DPV + l31) * 16;
    auto read_k = [&](half8_t (&kf)[2][Cfg::NKS], int slot) __attribute__((always_inline)) {
        const char* kb_ = smem + slot * Cfg::TILE + koff;
#pragma unroll
        for (int ks = 0; ks < Cfg::NKS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                kf[kb][ks] = *reinterpret_cast<const half8_t*>(kb_ + (ks * 128 + kb * 32) * 16);
    };

    floatx16 s[QB][2];  // S^T of the tile whose softmax comes next
    auto qk = [&](half8_t (&kf)[2][Cfg::NKS]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[j][0][r] = Cfg::MCOL ? 0.f : negm[j][r];
                s[j][1][r] = Cfg::MCOL ? 0.f : negm[j][r];
            }
#pragma unroll
        for (int ks = 0; ks < Cfg::NKS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    s[j][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[j][ks], s[j][kb], 0, 0, 0);
                }
    };

    // ---- prologue: packs 0 .. 3 in flight (pack p = step p-1, slot (p+3) & 3), packs 0 and 1 landed,
    // S^T of tile 0 computed
    stage(0, 3);
    stage(1, 0);
    if (nT > 1) stage(2, 1);
    if (nT > 2) stage(3, 2);
    wait_barrier(nT > 2 ? 2 : (nT > 1 ? 1 : 0));
    {
        half8_t kf[2][Cfg::NKS];
        read_k(kf, 3);
        qk(kf);
    }
    __builtin_amdgcn_sched_barrier(0);

    const int need_diag = diag_bias_log2 != 0.f ? 1 : 0;

    // Step u: softmax of tile u, O^T += V^T(u) P^T(u), S^T(u+1) = K(u+1) Q^T.  One loop body; what differs
    // between tiles and waves is three wave-uniform (scalar-branch) passes in front of the exponentials:
    //  * fix   : per-element fix-ups -- padded keys of the last tile, diagonal bias (then on every tile);
    //  * search: running-max search + deferred rescale.  Dropped (nomax) after tile 0 has anchored m_run when
    //            `qbound` proves that no exponent argument can exceed NOMAX_THR: P is then at most
    //            2^NOMAX_THR, inside fp16 range, and the row sum normalises it exactly as before;
    //  * exact : the wave keeps Q unscaled: scores are multiplied by c in fp32 before the exponential.
    // LAST = the final tile: no S^T to compute for a next one.
    // FAST = the common case as an instantiation of its own: no fix-ups (only the max search and the exact-scale
    // pass keep their scalar branches), and (u + 3 < nT) so that the ring handling is unconditional.  (A taken scalar branch costs a wave an
    // instruction-fetch bubble; the generic body skips over its rare passes with a dozen of them per tile.)
    int nomax = 0;
    auto step = [&](int u, auto last_c, auto fast_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        constexpr bool FAST = decltype(fast_c)::value;
        const int slot = u & 3;
        const int fix = FAST ? 0 : (LAST ? 1 : need_diag);
        const int search = !nomax;
        const int exact = !folded;
        // barrier u: pack u+2 (step u+1) has landed for everyone; its predecessor's slot takes pack u+4
        auto ring_sync = [&]() __attribute__((always_inline)) {
            if (FAST) {
                ring_wait_barrier<NPW_LO>();  // (waves with an extra piece per pack wait for one piece more)
                stage(u + 4, (u + 3) & 3);
            } else {
                wait_barrier(u + 2 < nT ? 1 : 0);
                if (u + 3 < nT) stage(u + 4, (u + 3) & 3);
            }
        };
        if (grpB) ring_sync();
        __builtin_amdgcn_sched_barrier(0);

        // ---- V^T fragments of tile u -> registers (landed one barrier ago), in flight under the softmax
        half8_t vf[4][Cfg::NDB];
        {
            const char* vb_ = smem + slot * Cfg::TILE + voff;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                for (int db = 0; db < Cfg::NDB; ++db)
                    vf[kc][db] = *reinterpret_cast<const half8_t*>(vb_ + (kc * 2 * Cfg::DPV + db * 32) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- online softmax, one query per lane; the packed P registers are the PV B operands
        half8_t pf[QB][4];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            if (fix) {
                const int qr = qrow0 + 32 * j;
                int kbase = u * 64 + 4 * hi;
                asm volatile("" : "+v"(kbase));  // keeps the index arithmetic of this rare pass inside its branch
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key0 = kbase + (r & 3) + 8 * (r >> 2);
                    if (need_diag && key0 == qr) s[j][0][r] += diag_u;
                    if (need_diag && key0 + 32 == qr) s[j][1][r] += diag_u;
                    if (key0 >= M) s[j][0][r] = -1e30f;
                    if (key0 + 32 >= M) s[j][1][r] = -1e30f;
                }
            }
            // s = exponent argument (units u) relative to m_run; the reference point moves (and O, l are
            // rescaled) only when the tile max exceeds it by more than RESCALE_THR -- or on tile 0, which
            // anchors it at the row's first-tile max.
            if (search) {
                float mt = fmaxf(s[j][0][0], s[j][1][0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, s[j][0][r]), s[j][1][r]);  // v_max3_f32
                {   // the other half of the wave holds the query's other 32 keys of the tile: v_permlane32_swap (VALU) instead
                    // of a ds_bpermute through the LDS pipe (the builtin inserts the wait state the swap needs)
                    const unsigned mb = __builtin_bit_cast(unsigned, mt);
                    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                    mt = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                }
                if (u == 0 || __builtin_amdgcn_readfirstlane((int)__any(mt > resc_thr)) != 0) {
                    float delta = (u == 0) ? mt : fmaxf(mt, 0.f);
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
                    const float alpha = __builtin_amdgcn_exp2f(-delta * cmul);
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
            }
            if (exact) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[j][0][r] *= cmul;
                    s[j][1][r] *= cmul;
                }
            }
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(s[j][kb][r]);
                    const float p1 = __builtin_amdgcn_exp2f(s[j][kb][r + 1]);
                    if (!Cfg::ONES) psum += p0 + p1;
                    pf[j][kb * 2 + (r >> 3)][r & 7] = (half_t)p0;
                    pf[j][kb * 2 + (r >> 3)][(r & 7) + 1] = (half_t)p1;
                }
            if (!Cfg::ONES) l_run[j] += psum;
            // (pins the exponentials in front of group A's barrier below: being free of side effects they would
            // otherwise be sunk behind the inline-asm barrier, into the MFMA block)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) asm volatile("" : "+v"(pf[j][kc]));
        }
        __builtin_amdgcn_sched_barrier(0);

        if (!grpB) ring_sync();
        __builtin_amdgcn_sched_barrier(0);

        // ---- K fragments of tile u+1 (same pack), in flight under the PV MFMAs
        half8_t kf[2][Cfg::NKS];
        if (!LAST) read_k(kf, slot);
        __builtin_amdgcn_sched_barrier(0);

        // The MFMA block runs at raised priority: against a partner wave in its softmax, an MFMA wave that loses the
        // issue arbitration (it does when it is the younger one) leaves the matrix pipe idle between MFMAs
        // (tools/ubench_rates.hip: 28 MFMAs beside a prioritised exp/cvt stream take 1590 cycles instead of 900).
        __builtin_amdgcn_s_setprio(1);
        // ---- O^T += V^T P^T  (row D of V^T is all ones when it is spare: O^T[D] = row sum)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int db = 0; db < Cfg::NDB; ++db)
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    o[j][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kc][db], pf[j][kc], o[j][db], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
        // ---- S^T of tile u+1
        if (!LAST) qk(kf);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const std::integral_constant<bool, true> yes;
    const std::integral_constant<bool, false> no_last, no;
    const std::integral_constant<bool, true> fast;
    {
        int u = 0;
        if (nT > 1) {
            step(0, no_last, no);
            u = 1;
            if (!need_diag) {
                bool safe = true;
#pragma unroll
                for (int j = 0; j < QB; ++j) safe = safe && (cmul * (qbound[j] - m_run[j]) <= NOMAX_THR);
                nomax = __builtin_amdgcn_readfirstlane((int)__all(safe));
            }
            if (!need_diag)
                for (; u + 3 < nT; ++u) step(u, no_last, fast);
            for (; u < nT - 1; ++u) step(u, no_last, no);
        }
        nomax = 0;  // the last tile has padded keys at -1e30: its maximum must be looked at
        step(u, yes, no);
    }

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
        // A row's 8-column groups sit split over the two half-waves (lane l31: columns 8k .. 8k+3, lane
        // l31 + 32: 8k+4 .. 8k+7).  One v_permlane32_swap per dword of a PAIR of groups leaves lanes 0-31 with the 16
        // contiguous bytes of group k and lanes 32-63 with those of group k+1: one 16-byte store per pair instead of
        // two 8-byte ones (the store tail of a row-per-lane epilogue is bound by store instructions, not bytes).
        {
            half_t* op = out + ((int64_t)b * Lq + (qr < Lq ? qr : 0)) * C + h * D;
#pragma unroll
            for (int db = 0; db < Cfg::NDB; ++db)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const int dA = db * 32 + gp * 16;  // first column of the pair
                    if (dA >= D) continue;
                    half4_t wa, wb;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        wa[jj] = (half_t)(o[j][db][(2 * gp) * 4 + jj] * inv);
                        wb[jj] = (half_t)(o[j][db][(2 * gp + 1) * 4 + jj] * inv);
                    }
                    if (dA + 8 < D) {
                        const u32x2 a = __builtin_bit_cast(u32x2, wa), bb = __builtin_bit_cast(u32x2, wb);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a[0], bb[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(a[1], bb[1], false, false);
                        u32x4 st;
                        st[0] = s0[0]; st[1] = s1[0]; st[2] = s0[1]; st[3] = s1[1];
                        if (qr < Lq) *reinterpret_cast<u32x4*>(op + dA + hi * 8) = st;
                    } else if (qr < Lq) {  // a lone 8-column group (D % 16 == 8): the two 8-byte halves as before
                        *reinterpret_cast<half4_t*>(op + dA + hi * 4) = wa;
                    }
                }
        }
    }
}

template <int D, int QB>
static void launch_flash(const half_t* q, const char* img, half_t* out, int B, int H, int Lq, int M, int nT,
                         int n_groups, float scale, float diag_bias, int64_t q_ld, const float* ktmax,
                         hipStream_t st) {
    using Cfg = AttnCfg<D>;
    // (per device and cheap: set on every launch rather than cached in a process-global flag)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_flash_kernel<D, QB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    const int nQblk = (Lq + 256 * QB - 1) / (256 * QB);
    const float log2e = 1.4426950408889634f;
    ProfScope ps(FRESCO_PROF_ATTN_FLASH, B * H, Lq, M, D, st);
    const int grid = H * nQblk * B;
    hipLaunchKernelGGL((attn_flash_kernel<D, QB>), dim3(grid), dim3(512), Cfg::LDS_BYTES, st, q, img,
                       ktmax, out, B, H, Lq, M, nT, B / n_groups, scale * log2e, diag_bias * log2e, q_ld);
}

template <int D>
static int launch_attn(const half_t* q, const half_t* k, const half_t* v, const int32_t* kv_rows,
                       half_t* out, char* ws, int B, int H, int Lq, int n_groups, int M,
                       int64_t group_rows, float scale, float diag_bias, int64_t q_ld, int64_t kv_ld,
                       hipStream_t st) {
    using Cfg = AttnCfg<D>;
    const int nT = ntiles_of(M);
    char* img = ws;
    float* ktmax = reinterpret_cast<float*>(ws + align_up((size_t)n_groups * H * (nT + 1) * Cfg::TILE, 256));
    dim3 pg(nT, H, n_groups);
    {
        ProfScope ps(FRESCO_PROF_KV_PACK, n_groups, H, M, D, st);
        hipLaunchKernelGGL((kv_pack_kernel<D>), pg, dim3(256), 0, st, k, v, kv_rows, img, ktmax, H, M, nT, group_rows,
                           kv_ld);
    }
    // Two query blocks (64 rows) per wave while the accumulators leave room (two waves per SIMD = 256
    // registers each): every K / V^T fragment read from LDS then feeds two (four) MFMAs.  Exception: a launch whose
    // 512-row workgroups would leave CUs idle (a frame shard of a multi-GPU run: 2 batch rows x 8 heads x 8 query blocks
    // = 128 workgroups for 256 CUs) takes 256-row workgroups instead -- twice as many, each half as long.
    if constexpr (D <= 48 && Cfg::MCOL) {
        const int grid2 = H * ((Lq + 511) / 512) * B;
        if (Lq > 256 && grid2 < device_cus())
            launch_flash<D, 1>(q, img, out, B, H, Lq, M, nT, n_groups, scale, diag_bias, q_ld, ktmax, st);
        else
            launch_flash<D, 2>(q, img, out, B, H, Lq, M, nT, n_groups, scale, diag_bias, q_ld, ktmax, st);
    } else {
        launch_flash<D, 1>(q, img, out, B, H, Lq, M, nT, n_groups, scale, diag_bias, q_ld, ktmax, st);
    }
    return check_launch();
}

static size_t attn_ws_bytes(int n_groups, int H, int M, int D) {
    const size_t nT = ntiles_of(M);
    const size_t dpk = (D + 15) / 16 * 16, dpv = (D + 31) / 32 * 32;
    return align_up((size_t)n_groups * H * (nT + 1) * ((dpk + dpv) * 128), 256) +
           align_up((size_t)n_groups * H * nT * sizeof(float), 256);
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

namespace fresco {
template <int KIN, int D>
static int launch_kvproj_attn(const half_t* q, const half_t* x, int64_t x_ld, const int32_t* x_rows, const half_t* Wk,
                              const half_t* Wv, half_t* out, char* ws, int B, int H, int Lq, int n_groups, int M,
                              float scale, int64_t q_ld, hipStream_t st) {
    using Cfg = AttnCfg<D>;
    const int nT = ntiles_of(M);
    char* img = ws;
    float* ktmax = reinterpret_cast<float*>(ws + align_up((size_t)n_groups * H * (nT + 1) * Cfg::TILE, 256));
    {
        ProfScope ps(FRESCO_PROF_KV_PACK, n_groups, H, M, -D, st);  // (d < 0: the fused projection + pack launch)
        constexpr int lds = KvProjCfg<KIN, D>::LDS_BYTES;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kvproj_pack_kernel<KIN, D>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((kvproj_pack_kernel<KIN, D>), dim3(nT, H * D / 160, n_groups), dim3(320), lds, st, x, x_ld, x_rows,
                           Wk, Wv, img, ktmax, H, M, nT);
    }
    if constexpr (D <= 48 && Cfg::MCOL) {
        const int grid2 = H * ((Lq + 511) / 512) * B;
        if (Lq > 256 && grid2 < device_cus())
            launch_flash<D, 1>(q, img, out, B, H, Lq, M, nT, n_groups, scale, 0.f, q_ld, ktmax, st);
        else
            launch_flash<D, 2>(q, img, out, B, H, Lq, M, nT, n_groups, scale, 0.f, q_ld, ktmax, st);
    } else {
        launch_flash<D, 1>(q, img, out, B, H, Lq, M, nT, n_groups, scale, 0.f, q_ld, ktmax, st);
    }
    return check_launch();
}
}  // namespace fresco

extern "C" int fresco_attn_kvproj_supported(int H, int D, int K_in) {
    return (H > 0 && (int64_t)H * D == K_in && ((D == 40 && K_in == 320) || (D == 80 && K_in == 640))) ? 1 : 0;
}

extern "C" int fresco_attn_fwd_kvproj(const void* q, const void* x, int64_t x_ld, const int32_t* x_rows, const void* Wk,
                                      const void* Wv, void* out, void* workspace, size_t workspace_bytes, int B, int H,
                                      int Lq, int D, int n_groups, int M, int K_in, float scale, int64_t q_ld,
                                      void* stream) {
    if (!q || !x || !x_rows || !Wk || !Wv || !out || !workspace) return FRESCO_EINVAL;
    if (B <= 0 || H <= 0 || Lq <= 0 || D <= 0 || n_groups <= 0 || M <= 0 || K_in <= 0) return FRESCO_EINVAL;
    if (B % n_groups != 0 || !(scale > 0.f)) return FRESCO_EINVAL;
    if (q_ld < (int64_t)H * D || q_ld % 8 != 0 || x_ld < K_in || x_ld % 8 != 0) return FRESCO_EINVAL;
    if (!fresco_attn_kvproj_supported(H, D, K_in)) return FRESCO_EUNSUPPORTED;
    if (workspace_bytes < attn_ws_bytes(n_groups, H, M, D)) return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const half_t* qh = static_cast<const half_t*>(q);
    const half_t* xh = static_cast<const half_t*>(x);
    const half_t* wk = static_cast<const half_t*>(Wk);
    const half_t* wv = static_cast<const half_t*>(Wv);
    half_t* oh = static_cast<half_t*>(out);
    char* ws = static_cast<char*>(workspace);
    if (D == 40)
        return launch_kvproj_attn<320, 40>(qh, xh, x_ld, x_rows, wk, wv, oh, ws, B, H, Lq, n_groups, M, scale, q_ld, st);
    return launch_kvproj_attn<640, 80>(qh, xh, x_ld, x_rows, wk, wv, oh, ws, B, H, Lq, n_groups, M, scale, q_ld, st);
}

extern "C" int fresco_attn_fwd(const void* q, const void* k, const void* v, const int32_t* kv_rows,
                               void* out, void* workspace, size_t workspace_bytes, int B, int H,
                               int Lq, int D, int n_groups, int M, int64_t group_rows, float scale,
                               float diag_bias, void* stream) {
    return fresco_attn_fwd_ld(q, k, v, kv_rows, out, workspace, workspace_bytes, B, H, Lq, D, n_groups, M,
                              group_rows, scale, diag_bias, (int64_t)H * D, (int64_t)H * D, stream);
}
