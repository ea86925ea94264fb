// Device-side pieces shared by opt.hip (generic kernels + host side) and opt_fast.hip (the 4-launch pipeline the
// SD-1.5 shapes run).  Not part of the public ABI.
#pragma once
#include "common.h"
#include <math.h>

namespace fresco {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// shared bilinear tap helper (same arithmetic as warp.hip: geometry.py:50-55,65-72)
// ------------------------------------------------------------------------------------------------
struct OTaps {
    int idx[4];
    float w[4];
};

__device__ __forceinline__ OTaps otaps(float fx, float fy, int x, int y, int h, int w) {
    const float gx = 2.f * ((float)x + fx) / (float)(w - 1) - 1.f;
    const float gy = 2.f * ((float)y + fy) / (float)(h - 1) - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f;
    const float x0c = fminf(fmaxf(x0f, -2.f), (float)w + 1.f);
    const float y0c = fminf(fmaxf(y0f, -2.f), (float)h + 1.f);
    const int x0 = (int)x0c, y0 = (int)y0c, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < w && x0f == x0c, vx1 = x1 >= 0 && x1 < w && x0f == x0c;
    const bool vy0 = y0 >= 0 && y0 < h && y0f == y0c, vy1 = y1 >= 0 && y1 < h && y0f == y0c;
    const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
    const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
    OTaps t;
    t.idx[0] = cy0 * w + cx0;
    t.idx[1] = cy0 * w + cx1;
    t.idx[2] = cy1 * w + cx0;
    t.idx[3] = cy1 * w + cx1;
    t.w[0] = (vx0 && vy0) ? (1.f - tx) * (1.f - ty) : 0.f;
    t.w[1] = (vx1 && vy0) ? tx * (1.f - ty) : 0.f;
    t.w[2] = (vx0 && vy1) ? (1.f - tx) * ty : 0.f;
    t.w[3] = (vx1 && vy1) ? tx * ty : 0.f;
    return t;
}

__device__ __forceinline__ float osample(const float* __restrict__ plane, const OTaps& t) {
    return plane[t.idx[0]] * t.w[0] + plane[t.idx[1]] * t.w[1] + plane[t.idx[2]] * t.w[2] +
           plane[t.idx[3]] * t.w[3];
}

__device__ __forceinline__ int sgn(float x) { return (x > 0.f) - (x < 0.f); }

constexpr int OCPT = 8;  // channels per thread in the temporal kernels

// Frame layout of the temporal term.  Single GPU: the n_loc = N frames of a CFG half form a ring,
// pair j = (frame j, frame (j+1) % N), n_pairs = N.  Frame-sharded (multi-GPU): the rank owns n_loc
// consecutive frames and receives the frame before (halo_l) and after (halo_r) them each iteration;
// slots 0 .. n_loc+1 = halo_l, local frames, halo_r; pair j = (slot j, slot j+1), n_pairs = n_loc + 1
// (the pair straddling the left boundary is evaluated redundantly by both neighbours).
struct TLayout {
    int n_loc, n_pairs, circular;
    const float* halo_l;  // (chunk, C, hw)
    const float* halo_r;
};

__device__ __forceinline__ const float* frame_plane(const float* cs, const TLayout& L, int ck, int slot, int c, int C,
                                                    int hw) {
    if (L.circular) return cs + ((int64_t)(ck * L.n_loc + slot) * C + c) * hw;
    if (slot == 0) return L.halo_l + ((int64_t)ck * C + c) * hw;
    if (slot == L.n_loc + 1) return L.halo_r + ((int64_t)ck * C + c) * hw;
    return cs + ((int64_t)(ck * L.n_loc + slot - 1) * C + c) * hw;
}

// For local frame fl with pairs  jf = the pair whose FIRST frame it is, jp = the pair whose SECOND:
// grad[fl][c][p] = k mf[jf][p] sgn2[jf] + k mb[jp][p] sgn1[jp] - sum_rowB[jf][p] w*sgn1[jf][src]
//                                                           - sum_rowF[jp][p] w*sgn2[jp][src]
// The two CSR rows of a pixel are shared by all channels: their first TG_MAXE entries are held in
// registers (a smooth flow gives ~4 entries per row), longer rows continue from memory.  (4 instead of 6 cached entries:
// -0.4 ms per config-3 step on the bench's near-uniform flows, whose rows have exactly 4 -- and a divergent tail loop for
// every longer row of a real flow field: measured, not kept.)
constexpr int TG_MAXE = 6;

// Per-thread state of the temporal gradient of pixel p of local frame (ck, fl): everything that is shared by the
// channels (pair indices, occlusion factors, the register-cached heads of the two CSR rows).
struct TGradArgs {
    const int8_t* sgn1;
    const int8_t* sgn2;
    const float* bwd_occ;
    const float* fwd_occ;
    const int* rowptr;
    const int* src;
    const float* wgt;
    TLayout L;
    float kscale;
};

struct TGradPixel {
    int bf, bp, bB, eB, bF, eF;
    float a1, a2;
    const int *sB, *sF;
    const float *wB, *wF;
    int iB[TG_MAXE], iF[TG_MAXE];
    float vB[TG_MAXE], vF[TG_MAXE];

    __device__ __forceinline__ void init(const TGradArgs& t, int b, int p, int hw) {
        const TLayout& L = t.L;
        const int ck = b / L.n_loc, fl = b % L.n_loc;
        const int NP = L.n_pairs;
        const int jf = L.circular ? fl : fl + 1;
        const int jp = L.circular ? (fl + L.n_loc - 1) % L.n_loc : fl;
        bf = ck * NP + jf;
        bp = ck * NP + jp;
        a2 = t.kscale * (1.f - t.fwd_occ[(int64_t)jf * hw + p]);
        a1 = t.kscale * (1.f - t.bwd_occ[(int64_t)jp * hw + p]);
        const int* rpB = t.rowptr + (int64_t)(0 * NP + jf) * (hw + 1);
        const int* rpF = t.rowptr + (int64_t)(1 * NP + jp) * (hw + 1);
        bB = rpB[p], eB = rpB[p + 1];
        bF = rpF[p], eF = rpF[p + 1];
        sB = t.src + (int64_t)(0 * NP + jf) * 4 * hw;
        wB = t.wgt + (int64_t)(0 * NP + jf) * 4 * hw;
        sF = t.src + (int64_t)(1 * NP + jp) * 4 * hw;
        wF = t.wgt + (int64_t)(1 * NP + jp) * 4 * hw;
#pragma unroll
        for (int e = 0; e < TG_MAXE; ++e) {
            const bool okB = bB + e < eB, okF = bF + e < eF;
            iB[e] = okB ? sB[bB + e] : 0;  // weight 0 -> the (valid) index 0 contributes nothing
            vB[e] = okB ? wB[bB + e] : 0.f;
            iF[e] = okF ? sF[bF + e] : 0;
            vF[e] = okF ? wF[bF + e] : 0.f;
        }
    }
    // gradients of the 8 channels of octet c8 at the pixel (signs: [pair][C/8][hw][8] bytes, one 8-byte word per load)
    __device__ __forceinline__ void values(const TGradArgs& t, int c8, int p, int C8, int hw, float (&out)[8]) const {
        const uint64_t* s1f = reinterpret_cast<const uint64_t*>(t.sgn1) + ((int64_t)bf * C8 + c8) * hw;
        const uint64_t* s2f = reinterpret_cast<const uint64_t*>(t.sgn2) + ((int64_t)bf * C8 + c8) * hw;
        const uint64_t* s1p = reinterpret_cast<const uint64_t*>(t.sgn1) + ((int64_t)bp * C8 + c8) * hw;
        const uint64_t* s2p = reinterpret_cast<const uint64_t*>(t.sgn2) + ((int64_t)bp * C8 + c8) * hw;
        auto sg = [](uint64_t w, int k) { return (float)(int8_t)(uint8_t)(w >> (8 * k)); };
        const uint64_t d2 = s2f[p], d1 = s1p[p];
        uint64_t gB[TG_MAXE], gF[TG_MAXE];
#pragma unroll
        for (int e = 0; e < TG_MAXE; ++e) {
            gB[e] = s1f[iB[e]];
            gF[e] = s2p[iF[e]];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float adj = 0.f, adj2 = 0.f;
#pragma unroll
            for (int e = 0; e < TG_MAXE; ++e) {
                adj = fmaf(vB[e], sg(gB[e], k), adj);
                adj2 = fmaf(vF[e], sg(gF[e], k), adj2);
            }
            out[k] = a2 * sg(d2, k) + a1 * sg(d1, k) - adj - adj2;
        }
        for (int e = bB + TG_MAXE; e < eB; ++e) {  // rows longer than the register cache (rare)
            const uint64_t g = s1f[sB[e]];
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] -= wB[e] * sg(g, k);
        }
        for (int e = bF + TG_MAXE; e < eF; ++e) {
            const uint64_t g = s2p[sF[e]];
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] -= wF[e] * sg(g, k);
        }
    }
};

// sign(G - T) is stored as ONE BYTE = the high byte of the fp16 value of the sign (0x3C: +1, 0xBC: -1, 0x00: 0), so
// that the fp16-MFMA kernel expands four of them to packed halfs with two v_perm_b32 (a plain int8 sign costs ~4 VALU
// operations per value there, enough to make the S V kernel issue-bound next to its MFMAs).
__device__ __forceinline__ int8_t sign_byte(float d) { return (int8_t)(d > 0.f ? 0x3C : (d < 0.f ? 0xBC : 0)); }
__device__ __forceinline__ float sign_from_byte(uint32_t b) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)((b & 0xffu) << 8));
}

// (the condition under which sv16b_kernel runs: its operands -- vh / vl and the sign bytes -- are then stored pre-tiled:
// V as [plane][channel tile of 128][pixel chunk of 32][128][32] halfs, S as [plane][pixel tile of 256][chunk of 32][256][32] bytes,
// both with XOR-swizzled units inside a row: see sv16b_kernel)
__host__ __device__ __forceinline__ bool sv_tiled_layout(int hw, int C) { return hw % 256 == 0 && C % 128 == 0; }
// (the condition under which gram16y_kernel runs: the pixel-major operand copies are then stored pre-tiled AND swizzled:
// [plane][pixel tile of 128][channel chunk of 16][128 pixels][2 x 16-byte units], the two units of pixel row r swapped
// when (r >> 3) & 1 -- the image a linear LDS-DMA copy needs for conflict-free ds_read_b128 on 32-byte rows)
__host__ __device__ __forceinline__ bool gram_x_layout(int hw, int C, int min_hw = 512) {
    return hw % 256 == 0 && hw >= min_hw && C % 32 == 0;
}

// ---- host side ---------------------------------------------------------------------------------------------------
struct AdamArgs {
    float beta1, beta2, step_size, bc2_sqrt, eps;
};

// the caller-provided workspace of one optimize_feature call, carved by opt_ws_layout (opt.hip)
struct OptWs {
    float *grad, *m, *v, *vt, *dvt, *nrm, *wgt, *part, *dotp;
    float* gpart;  // planes <= 64 pixels: the 8 split-K partial tiles of the Gram product per plane (gram16sp / gram16sr)
    half_t *vh, *vl, *vph, *vpl;
    int8_t *sgn1, *sgn2, *ssign;
    int *rowptr, *cursor, *src;
};

// opt_fast.hip: the pipeline the SD-1.5 shapes run (hw % 64 == 0, C % 8 == 0, a Gram target): per Adam iteration
// prep -> gram -> S V -> adam, four launches.
bool opt_fast_ok(int C, int h, int w, int has_s);
// partial sums of squares of the initial features (the later ones come out of the Adam kernel); once per call.
// Bg = planes of the WHOLE batch (fixes the work split, i.e. the order of the partial sums)
void opt_fast_begin(const OptWs& w, const float* cs, int planes, int C, int hw, int Bg, hipStream_t st);
// one closure evaluation on `nck` CFG halves starting at the pointers given (w, cs, target already offset to the first
// half); mode 0 = Adam step, mode 1 = write the gradient to gout.  Bg = global batch (normalises both loss terms);
// sync (optional): events that order this pipeline against a second one on another stream
struct FastSync {
    hipEvent_t wait_before_gram = nullptr, record_after_gram = nullptr, record_after_sv = nullptr, wait_before_adam = nullptr;
    int parts = 3;  // 1: prep + gram + S V, 2: adam (a closure may be issued in two host calls)
    // frame-sharded form: part 1 evaluates only what needs no halo frame, part 2 starts with the signs of the two pairs
    // that do (PrepArgs::phase 1 / 2) -- the neighbour exchange of the halo frames then runs under Gram + S V
    int halo_split = 0;
};
void opt_fast_closure(const OptWs& w, float* cs, const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                      const float* bwd_occ, const float* target, int nck, int C, int h, int wd, float intra_weight,
                      int has_t, int mode, float* gout, float* loss, AdamArgs a, hipStream_t st, const TLayout& L,
                      int Bg, const FastSync* sync = nullptr);
// the S V product on split-fp16 MFMAs for plain (un-tiled) operand layouts; dotp (optional): per-(128-channel tile)
// partial sums of <V, dV> per pixel
void launch_sv16_plain(const half_t* vh, const half_t* vl, const int8_t* ssign, float* dvt, float* dotp, int B, int C,
                       int hw, float alpha, hipStream_t st);
// opt.hip: the generic split-fp16 Gram kernel (plain layouts, any hw, C % 8 == 0)
void launch_gram16_plain(const half_t* vph, const half_t* vpl, const float* target, int8_t* ssign, float* loss, int B,
                         int C, int hw, hipStream_t st);

}  // namespace fresco
