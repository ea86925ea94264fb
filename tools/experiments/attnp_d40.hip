// Software-pipelined flash attention for the efficient cross-frame pass at the decoder's head dim 40
// (reference: src/diffusion_hacked.py:225-247, 303-305; same arithmetic contract as attn_flash_kernel of attn.hip,
// same packed key images, with the V^T half of a pack lagging its K half by TWO tiles).
//
// Why a second kernel.  attn_flash_kernel runs two waves per SIMD that alternate whole softmax and MFMA blocks
// behind one barrier; measured, that ping-pong overlaps only about half of the softmax with the partner's matrix
// work (matrix pipe 63 % busy, profiles/r02_pmc_attn_v2.csv) and none of the 18 variants measured at the start of
// round 3 moved it by more than 1 % (profiles/r03_ab_variants.txt).  Here ONE wave owns a SIMD (4 waves per
// workgroup, 64 query rows each, up to 512 registers) and overlaps the two pipes inside its own instruction stream:
// in step t the wave issues
//     PV(t-1):  O^T += V^T(t-1) P^T(t-1)      16 MFMAs      (P of the previous step)
//     QK(t+1):  S^T(t+1) = K(t+1) Q^T         12 MFMAs      (scores for the next step)
//     softmax(t): P(t) = exp2(S(t)), packed   64 v_exp_f32 + 32 v_cvt_pk_f16_f32
// as ONE hand-ordered stream: after every MFMA the 3-4 vector instructions that fit its 32-cycle shadow, pinned by
// sched_barrier (the order below IS the schedule: hipcc only allocates registers and inserts waits).  The three
// strands touch disjoint registers (S and P are double-buffered and swap roles every step, the loop is unrolled by
// two), so nothing in a step waits for anything of the same step.
//   * MFMAs are inline asm so that their operands can be placed by register class: accumulators O^T (64), the
//     resident Q fragments (24) and the K / V^T fragments (56; ds_read_b128 straight into AGPRs) live in the
//     accumulator file, S (2 x 64) and P (2 x 32) in the VGPRs the vector ALU can reach: ~350 registers, no copies.
//     hipcc knows nothing about an asm MFMA, so the distances its hazard rules would enforce are kept by the schedule
//     itself (a result is read by the vector ALU at least three MFMAs after the one that wrote it) or by explicit
//     s_nop where a rare pass reads accumulators.
//   * key packs arrive by LDS-DMA into a 4-slot ring: pack p = K(p) || V^T(p-2) is everything step p-1 reads; at the
//     barrier of step t pack t+2 has landed (the V^T fragments of the NEXT step are read under the QK MFMAs of this
//     one), pack t+3 is in flight and pack t+4 is requested, its pieces spread over the step.
//   * the rare passes (running-max search + deferred rescale, padded keys of the last tile) run in a sequential
//     "generic" step [PV(t-1) | softmax(t) | QK(t+1)] that shares the data flow: tile 0, the last three tiles, and
//     every tile of waves whose logit bound does not rule out fp16 overflow.  The pipelined loop comes in two
//     instantiations, scale folded into Q or applied per score (large-logit waves); per-wave choice, one ring protocol.
// MFMA 32x32x16 f16 operand layout: see attn.hip.
#include "attn_cfg.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace fresco {

namespace {

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// O^T += A B with the accumulator and the A fragment in the accumulator file, B (the packed P) in VGPRs.
// PAD: two wait states in front, inside the statement (generic steps: hipcc may copy an operand into the accumulator
// file, or finish writing P, immediately before an asm statement it does not recognise as an MFMA).
template <bool PAD = false>
__device__ __forceinline__ void mfma_pv(floatx16& acc, const half8_t& a, const u32x4& b) {
    if (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
}
// S^T = A B (first k-step: zero C operand) / S^T += A B, result in VGPRs, both fragments in the accumulator file
template <bool PAD = false>
__device__ __forceinline__ void mfma_qk0(floatx16& s, const half8_t& a, const half8_t& b) {
    if (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(s) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(s) : "a"(a), "a"(b));
}
template <bool PAD = false>
__device__ __forceinline__ void mfma_qk(floatx16& s, const half8_t& a, const half8_t& b) {
    if (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "a"(a), "a"(b));
}

// The vector-ALU strand of one pipelined step as a flat list of items, in issue order.  Unit u = two adjacent scores
// of one query: (exp, exp, cvt_pk), the cvt skewed one unit behind its exps; with the exact (unfolded) scale two
// multiplies run one unit ahead.  kind: 0 / 1 = exp of element 0 / 1, 2 = cvt, 3 / 4 = multiply of element 0 / 1.
struct VItem { int kind, unit; };
constexpr int CVT_SKEW = 3;  // a v_cvt_pk is issued this many units behind the v_exp pair it packs (trans latency)
template <bool EXACT, int NU_>
struct VSched {
    static constexpr int NU = NU_;
    static constexpr int NI = EXACT ? 5 * NU : 3 * NU;
    VItem it[NI];
    constexpr VSched() : it() {
        int n = 0;
        if (EXACT) { it[n++] = {3, 0}; it[n++] = {4, 0}; }
        for (int u = 0; u < NU; ++u) {
            if (EXACT && u + 1 < NU) { it[n++] = {3, u + 1}; it[n++] = {4, u + 1}; }
            it[n++] = {0, u};
            it[n++] = {1, u};
            if (u >= CVT_SKEW) it[n++] = {2, u - CVT_SKEW};
        }
        for (int u = NU - CVT_SKEW; u < NU; ++u) it[n++] = {2, u};
    }
};

}  // namespace

// grid (H * nQblk * B), PW waves x QB blocks of 32 query rows; blockIdx.x = (b * nQblk + qblk) * H + h.
// (QB, PW) = (1, 8): two waves per SIMD, <= 256 registers each -- the shipped form: the two instruction streams fill
// each other's issue gaps while the matrix pipe paces both; (2, 4): one wave per SIMD, every K / V^T fragment feeds two
// MFMAs, but ALL issue (96 vector instructions, 14 fragment reads, 4 DMA requests, waits and scalar code per 28 MFMAs)
// serialises in one stream: measured 407 us against 379 for attn_flash_kernel (profiles/r03_attn_pipe.txt).
template <int D, int QB, int PW>
__global__ __launch_bounds__(PW * 64, PW / 4) void attn_pipe_kernel(const half_t* __restrict__ q, const char* __restrict__ img,
                                                         const float* __restrict__ ktmax, half_t* __restrict__ out,
                                                         int B, int H, int Lq, int M, int nT, int batch_per_group,
                                                         float scale_log2, int64_t q_ld) {
    using Cfg = AttnCfg<D>;
    static_assert(Cfg::MCOL && Cfg::ONES, "the pipelined kernel relies on the spare K column / V^T row");
    static_assert(QB == 1 || QB == 2, "one or two query blocks per wave");
    constexpr int NKS = Cfg::NKS, NDB = Cfg::NDB, DPV = Cfg::DPV, TILE = Cfg::TILE;
    constexpr int ROWS = PW * 32 * QB;  // query rows per workgroup
    constexpr int NBUF = PIPE_NBUF, LEAD = PIPE_LEAD;  // ring slots; pack t + LEAD is requested in step t
    constexpr int NPV = 4 * NDB * QB, NQK = NKS * 2 * QB, NM = NPV + NQK;  // MFMAs per step
    extern __shared__ __attribute__((aligned(16))) char smem[];

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
    const int npk = nT + 2;  // packs 0 .. nT+1

    // ---- Q fragments (B operand of S^T = K Q^T) and |q|^2 ------------------------------------------------------
    half8_t qf[QB][NKS];
    float q2[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int qr = qrow0 + 32 * j;
        const half_t* qp = q + ((int64_t)b * Lq + (qr < Lq ? qr : Lq - 1)) * q_ld + h * D;
        q2[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            half8_t t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (d0 < D) t = *reinterpret_cast<const half8_t*>(qp + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) q2[j] = fmaf((float)t[e], (float)t[e], q2[j]);
            qf[j][ks] = t;
        }
        q2[j] += __shfl_xor(q2[j], 32, 64);
    }
    // Cauchy-Schwarz bound on the logits, the folded / exact scale decision and the accumulator units: as in
    // attn_flash_kernel (attn.hip), per wave
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
    const int folded = __builtin_amdgcn_readfirstlane((int)__all(fold_ok));
    const float qs = folded ? scale_log2 : 1.f;
    const float cmul = folded ? 1.f : scale_log2;
    float qbound[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        qbound[j] = qs * sqrtf(q2[j]) * kmax * 1.001f + 1e-3f;
        if (folded) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[j][ks][e] = (half_t)((float)qf[j][ks][e] * scale_log2);
        }
    }
    const float resc_thr = RESCALE_THR / cmul;

    // ---- ring: pack p -> slot p & 3.  Every wave moves 1/PW of a pack: NFULL whole 1 KiB pieces (one
    // global_load_lds_dwordx4 each) and, when the share is not whole KiB, one shorter piece issued with the lanes beyond
    // it masked off (EXEC is changed and restored inside the asm statement).  Every wave issues exactly NPW pieces per
    // pack, so the counted waits need no per-wave case and the pipelined step carries no branch.
    constexpr int SHARE = TILE / PW;              // bytes per wave and pack (3584 / 1792 at D = 40)
    static_assert(SHARE * PW == TILE && SHARE % 16 == 0, "a pack must split evenly over the waves");
    constexpr int NFULL = SHARE / 1024;           // whole pieces
    constexpr int REM_LANES = (SHARE % 1024) / 16;  // lanes of the short piece (0: none)
    constexpr int NPW = NFULL + (REM_LANES ? 1 : 0);  // pieces per wave and pack
    constexpr unsigned REM_LO = REM_LANES >= 32 ? 0xffffffffu : ((1u << REM_LANES) - 1u);
    constexpr unsigned REM_HI = REM_LANES > 32 ? ((1u << (REM_LANES - 32)) - 1u) : 0u;
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
    const char* src = img + (int64_t)(g * H + h) * npk * TILE + wave_s * SHARE;
    const uint32_t lane_off = lane * 16;
    auto stage_piece = [&](int p, int i) __attribute__((always_inline)) {  // this wave's piece i of pack p
        const char* sp = src + (int64_t)p * TILE + i * 1024;
        const uint32_t m0v = lds0 + (p & (NBUF - 1)) * TILE + wave_s * SHARE + i * 1024;
        if (i < NFULL) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sp), "s"(m0v)
                         : "memory");
        } else {
            uint64_t sav;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, %5\n\ts_mov_b32 m0, %3\n\t"
                         "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                         : "=&s"(sav)
                         : "v"(lane_off), "s"(sp), "s"(m0v), "n"(REM_LO), "n"(REM_HI)
                         : "memory");
        }
    };
    auto stage = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) stage_piece(p, i);
    };
    // barrier of step t: pack t+2 has landed for everyone (own pieces: counted vmcnt; the packs t+3 .. t+LEAD-1 requested
    // in earlier steps may stay in flight), all fragment reads of the slot that is refilled next are complete
    auto ring_wait = [&](int newer) __attribute__((always_inline)) {  // newer = packs after t+2 already requested
        static_assert(LEAD >= 4 && LEAD <= 7, "ring_wait enumerates LEAD - 3 <= 4 packs in flight");
        if (newer >= 4 && LEAD > 6) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(4 * NPW) : "memory");
        else if (newer >= 3 && LEAD > 5) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(3 * NPW) : "memory");
        else if (newer >= 2 && LEAD > 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NPW) : "memory");
        else if (newer >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // ---- state ------------------------------------------------------------------------------------------------
    floatx16 o[QB][NDB];
    floatx16 sA[QB][2], sB[QB][2];  // S^T of the tile whose softmax comes next / of the one after
    u32x4 pfA[QB][4], pfB[QB][4];   // packed P (8 halfs) of the newest / the previous tile
    half8_t kf[2][NKS], vf[4][NDB];
    float m_run[QB];
    constexpr int MKS = D / 16, MHI = (D % 16) / 8, ME = D % 8;  // where -m_run sits in Q's spare column
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m_run[j] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][db][r] = 0.f;
    }
    const int koff = (hi * 64 + l31) * 16;
    const int voff = Cfg::KTILE + (hi * DPV + l31) * 16;
    auto read_k = [&](const char* kb_) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) kf[kb][ks] = *reinterpret_cast<const half8_t*>(kb_ + (ks * 128 + kb * 32) * 16);
    };
    auto read_v = [&](const char* vb_) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int db = 0; db < NDB; ++db) vf[kc][db] = *reinterpret_cast<const half8_t*>(vb_ + (kc * 2 * DPV + db * 32) * 16);
    };
    // (generic steps: hipcc may copy an operand into the accumulator file, or finish writing P, right in front of an
    // asm MFMA it does not recognise as one: two wait states first)
    auto qk_all = [&](floatx16 (&s)[QB][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    if (ks == 0) mfma_qk0<true>(s[j][kb], kf[kb][ks], qf[j][ks]);
                    else mfma_qk<true>(s[j][kb], kf[kb][ks], qf[j][ks]);
                }
    };
    auto pv_all = [&](u32x4 (&pf)[QB][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    mfma_pv<true>(o[j][db], vf[kc][db], pf[j][kc]);
                }
    };
    // hipcc does not know that the asm statements above are MFMAs: before the vector ALU reads their results outside the
    // schedule's own distances, wait out the longest MFMA (the operands pin the reads behind this statement)
    auto settle_o = [&]() __attribute__((always_inline)) {
        static_assert(NDB == 2, "operand list of settle_o");
        if constexpr (QB == 2)
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[QB - 1][0]), "+a"(o[QB - 1][1]));
        else
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[0][0]), "+a"(o[0][1]));
    };
    auto settle_s = [&](floatx16 (&s)[QB][2]) __attribute__((always_inline)) {
        if constexpr (QB == 2)
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[QB - 1][0]), "+v"(s[QB - 1][1]));
        else
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s[0][0]), "+v"(s[0][1]));
    };

    // ---- prologue: packs 0 .. LEAD-1 requested (nT >= PIPE_MIN_TILES >= LEAD - 2), pack 0 landed, S^T(0) -------------
#pragma unroll
    for (int p = 0; p < LEAD; ++p) stage(p);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((LEAD - 1) * NPW) : "memory");
    read_k(smem + koff);
    qk_all(sA);
    settle_s(sA);

    // ---- generic step t: [PV(t-1) | softmax(t) with the rare passes | QK(t+1)], S in sA, P in pfB ------------------
    int nomax = 0;
    auto gstep = [&](int t, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        {   // packs requested so far: up to min(t + LEAD, npk) - 1; those after t+2 may stay in flight
            const int last = (t + LEAD < npk ? t + LEAD : npk) - 1;
            ring_wait(last - (t + 2));
        }
        if (t + LEAD < npk) stage(t + LEAD);
        const char* slot = smem + ((t + 1) & (NBUF - 1)) * TILE;
        if (!FIRST) {
            read_v(slot + voff);
            pv_all(pfB);
        }
        const int search = !nomax;
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            if (LAST) {  // padded keys of the last tile
                int kbase = t * 64 + 4 * hi;
                asm volatile("" : "+v"(kbase));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key0 = kbase + (r & 3) + 8 * (r >> 2);
                    if (key0 >= M) sA[j][0][r] = -1e30f;
                    if (key0 + 32 >= M) sA[j][1][r] = -1e30f;
                }
            }
            if (search) {
                float mt = fmaxf(sA[j][0][0], sA[j][1][0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) mt = fmaxf(fmaxf(mt, sA[j][0][r]), sA[j][1][r]);
                {
                    const unsigned mb = __builtin_bit_cast(unsigned, mt);
                    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                    mt = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                }
                if (FIRST || __builtin_amdgcn_readfirstlane((int)__any(mt > resc_thr)) != 0) {
                    float delta = FIRST ? mt : fmaxf(mt, 0.f);
                    // m_run stays on the fp16 grid (and finite): the value the MFMA subtracts is the one alpha is computed from
                    const float m_new = (float)(half_t)fminf(fmaxf(m_run[j] + delta, -6.0e4f), 6.0e4f);
                    delta = m_new - m_run[j];
                    m_run[j] = m_new;
                    const half_t nm = (half_t)(-m_new);
                    qf[j][MKS][ME] = (hi == MHI) ? nm : qf[j][MKS][ME];
                    const float alpha = __builtin_amdgcn_exp2f(-delta * cmul);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sA[j][0][r] -= delta;
                        sA[j][1][r] -= delta;
                    }
                    if (!FIRST) {
                        settle_o();
#pragma unroll
                        for (int db = 0; db < NDB; ++db)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[j][db][r] *= alpha;
                    }
                }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(sA[j][kb][r] * cmul);
                    const float p1 = __builtin_amdgcn_exp2f(sA[j][kb][r + 1] * cmul);
                    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                    half2_t pp;
                    pp[0] = (half_t)p0;
                    pp[1] = (half_t)p1;
                    pfB[j][kb * 2 + (r >> 3)][(r & 7) >> 1] = __builtin_bit_cast(uint32_t, pp);
                }
        }
        if (!LAST) {
            read_k(slot + koff);
            qk_all(sA);
        }
    };

    // ---- pipelined step t (no rare passes): MFMA g, then the vector items that fit its shadow.  Every instruction of
    // the two strands is an asm volatile statement: their source order IS the issue order (hipcc floats plain,
    // side-effect-free vector code past sched_barrier and sinks it below the MFMAs).  What hipcc still places: the
    // fragment ds_reads (with their lgkmcnt waits) and the scalar address arithmetic, held in place by sched_barrier.
    // Hazards the asm hides from hipcc, kept by construction: a v_cvt_pk reads its v_exp results two or more
    // instructions later (trans forwarding needs one), the vector ALU reads an S^T block >= 3 MFMAs after the MFMA that
    // completed it, P is read by MFMAs of the NEXT step only.
    half8_t qfa[QB][NKS];  // Q fragments of the pipelined loop: defined once in front of it, accumulator file
    const float cmul_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cmul)));
    auto fstep = [&](int t, floatx16 (&sCur)[QB][2], floatx16 (&sNew)[QB][2], u32x4 (&pfOld)[QB][4],
                     u32x4 (&pfNew)[QB][4], auto exact_c) __attribute__((always_inline)) {
        constexpr bool EXACT = decltype(exact_c)::value;
        constexpr VSched<EXACT, QB * 16> VS{};
        constexpr int NI = VSched<EXACT, QB * 16>::NI;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((LEAD - 3) * NPW) : "memory");
        const char* kb_ = smem + ((t + 1) & (NBUF - 1)) * TILE + koff;  // K(t+1): pack t+1
        const char* vb_ = smem + ((t + 2) & (NBUF - 1)) * TILE + voff;  // V^T(t): pack t+2, for the next step
        float pe[QB * 16][2], sm[QB * 16][2];
        static_for<NM>([&](auto gc) __attribute__((always_inline)) {
            constexpr int G = decltype(gc)::value;
            if constexpr (G < NPV) {
                constexpr int kc = G / (NDB * QB), db = (G / QB) % NDB, j = G % QB;
                mfma_pv(o[j][db], vf[kc][db], pfOld[j][kc]);
            } else {
                constexpr int x = G - NPV, ks = x / (2 * QB), kb = (x / QB) % 2, j = x % QB;
                if constexpr (ks == 0) mfma_qk0(sNew[j][kb], kf[kb][ks], qfa[j][ks]);
                else mfma_qk(sNew[j][kb], kf[kb][ks], qfa[j][ks]);
            }
            // fragment reads: K(t+1) under the first PV MFMAs, V^T(t) (for the next step) spread under the QK MFMAs
            if constexpr (G >= 1 && G < 1 + 2 * NKS) {
                constexpr int i = G - 1, ks = i / 2, kb = i % 2;
                kf[kb][ks] = *reinterpret_cast<const half8_t*>(kb_ + (ks * 128 + kb * 32) * 16);
            }
            if constexpr (G >= NPV) {
                static_for<4 * NDB>([&](auto vc) __attribute__((always_inline)) {
                    constexpr int i = decltype(vc)::value, kc = i / NDB, db = i % NDB;
                    if constexpr (NPV + i * NQK / (4 * NDB) == G)
                        vf[kc][db] = *reinterpret_cast<const half8_t*>(vb_ + (kc * 2 * DPV + db * 32) * 16);
                });
            }
            // pack t+LEAD -> a slot nobody reads any more, its NPW pieces spread over the step
            if constexpr (G % (NM / NPW) == 1 && G / (NM / NPW) < NPW) stage_piece(t + LEAD, G / (NM / NPW));
            // vector items [G*NI/NM, (G+1)*NI/NM)
            constexpr int i0 = G * NI / NM, i1 = (G + 1) * NI / NM;
            static_for<i1 - i0>([&](auto ic_) __attribute__((always_inline)) {
                constexpr int I = i0 + decltype(ic_)::value;
                constexpr int kind = VS.it[I].kind, u = VS.it[I].unit;
                constexpr int j = u / 16, kb = (u / 8) % 2, r = (u % 8) * 2;
                if constexpr (kind == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sm[u][0]) : "s"(cmul_s), "v"(sCur[j][kb][r]));
                if constexpr (kind == 4) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sm[u][1]) : "s"(cmul_s), "v"(sCur[j][kb][r + 1]));
                if constexpr (kind == 0) {
                    if constexpr (EXACT) asm volatile("v_exp_f32 %0, %1" : "=v"(pe[u][0]) : "v"(sm[u][0]));
                    else asm volatile("v_exp_f32 %0, %1" : "=v"(pe[u][0]) : "v"(sCur[j][kb][r]));
                }
                if constexpr (kind == 1) {
                    if constexpr (EXACT) asm volatile("v_exp_f32 %0, %1" : "=v"(pe[u][1]) : "v"(sm[u][1]));
                    else asm volatile("v_exp_f32 %0, %1" : "=v"(pe[u][1]) : "v"(sCur[j][kb][r + 1]));
                }
                if constexpr (kind == 2) {
                    uint32_t w;
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(pe[u][0]), "v"(pe[u][1]));
                    pfNew[j][kb * 2 + (r >> 3)][(r & 7) >> 1] = w;
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    const std::integral_constant<bool, true> yes;
    const std::integral_constant<bool, false> no;
    gstep(0, yes, no);
    {
        bool safe = true;
#pragma unroll
        for (int j = 0; j < QB; ++j) safe = safe && (cmul * (qbound[j] - m_run[j]) <= NOMAX_THR);
        nomax = __builtin_amdgcn_readfirstlane((int)__all(safe));
    }
    int t = 1;
    if (nomax) {
        // pipelined steps request pack t+LEAD unconditionally: pairs while (t + 1) + LEAD < npk = nT + 2
        if (t + LEAD <= nT) read_v(smem + ((t + 1) & (NBUF - 1)) * TILE + voff);  // V^T(t-1) for the first pipelined step
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                qfa[j][ks] = qf[j][ks];
                asm volatile("" : "+a"(qfa[j][ks]));
            }
        if (folded) {
            for (; t + LEAD <= nT; t += 2) {
                fstep(t, sA, sB, pfB, pfA, no);
                fstep(t + 1, sB, sA, pfA, pfB, no);
            }
        } else {
            for (; t + LEAD <= nT; t += 2) {
                fstep(t, sA, sB, pfB, pfA, yes);
                fstep(t + 1, sB, sA, pfA, pfB, yes);
            }
        }
    }
    for (; t < nT - 1; ++t) gstep(t, no, no);
    nomax = 0;  // the last tile has padded keys at -1e30: its maximum must be looked at
    gstep(t, no, yes);
    // final PV(nT-1): V^T(nT-1) sits in pack nT+1 (landed: the last barrier waited for everything)
    read_v(smem + ((nT + 1) & (NBUF - 1)) * TILE + voff);
    pv_all(pfB);
    settle_o();

    // ---- epilogue: normalise, store O[q][h*D + d] (16-byte stores through v_permlane32_swap pairs, as attn.hip) -----
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        constexpr int rr = D % 32;  // O^T row D = the row sum (ones row of V^T)
        const float l_tot = __shfl(o[j][D / 32][(rr & 3) + 4 * (rr >> 3)], l31 + 32 * ((rr >> 2) & 1), 64);
        const float inv = 1.f / l_tot;
        const int qr = qrow0 + 32 * j;
        half_t* op = out + ((int64_t)b * Lq + (qr < Lq ? qr : 0)) * C + h * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int dA = db * 32 + gp * 16;
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
                } else if (qr < Lq) {
                    *reinterpret_cast<half4_t*>(op + dA + hi * 4) = wa;
                }
            }
    }
}

// FRESCO_ATTN_PIPE selects the kernel behind fresco_attn_fwd at D = 40 without diagonal bias (read once; A/B timing of
// the kernels in one process image): 0 = attn_flash_kernel (attn.hip), 2 = the one-wave-per-SIMD form of this file,
// anything else / unset = the shipped two-waves-per-SIMD form.
static int pipe_mode() {
    static const int mode = [] {
        const char* e = getenv("FRESCO_ATTN_PIPE");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

bool attn_pipe_supported(int D, int nT, float diag_bias) {
    return pipe_mode() != 0 && D == 40 && nT >= PIPE_MIN_TILES && diag_bias == 0.f;
}

template <int D, int QB, int PW>
static void launch_pipe(const half_t* q, const char* img, const float* ktmax, half_t* out, int B, int H, int Lq, int M,
                        int nT, int n_groups, float scale, int64_t q_ld, hipStream_t st) {
    using Cfg = AttnCfg<D>;
    constexpr int LDS = PIPE_NBUF * Cfg::TILE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pipe_kernel<D, QB, PW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    constexpr int ROWS = PW * 32 * QB;
    const int nQblk = (Lq + ROWS - 1) / ROWS;
    const float log2e = 1.4426950408889634f;
    ProfScope ps(FRESCO_PROF_ATTN_FLASH, B * H, Lq, M, D, st);
    hipLaunchKernelGGL((attn_pipe_kernel<D, QB, PW>), dim3(H * nQblk * B), dim3(PW * 64), LDS, st, q, img,
                       ktmax, out, B, H, Lq, M, nT, B / n_groups, scale * log2e, q_ld);
}

int launch_attn_pipe(const half_t* q, const char* img, const float* ktmax, half_t* out, int B, int H, int Lq, int M,
                     int nT, int n_groups, float scale, int64_t q_ld, int D, hipStream_t st) {
    if (D != 40) return FRESCO_EUNSUPPORTED;
    if (pipe_mode() == 2)
        launch_pipe<40, 2, 4>(q, img, ktmax, out, B, H, Lq, M, nT, n_groups, scale, q_ld, st);
    else
        launch_pipe<40, 1, 8>(q, img, ktmax, out, B, H, Lq, M, nT, n_groups, scale, q_ld, st);
    return FRESCO_OK;
}

}  // namespace fresco
