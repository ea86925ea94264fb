// FRESCO feature optimisation (reference: src/diffusion_hacked.py:416-488): the pipeline the SD-1.5 shapes run.
// Everything here assumes hw % 64 == 0, C % 8 == 0 and a Gram target (opt_fast_ok); other shapes take the generic
// kernels of opt.hip.  One Adam iteration is FOUR launches, each reading and writing every tensor once:
//
//   prep   opt_prep_kernel   thread = (pixel, slice of 8K channels, temporal pair): |x[p]| from the partial sums of squares
//                            the previous Adam launch left behind; V = x/|x| written as fp16 hi + lo in both orientations
//                            (channel-major for S V, pixel-major for the Gram product, each in the tiling its reader's
//                            LDS-DMA wants); and, with the same x in registers, the int8 residual signs of the temporal term
//                            (4-tap gathers of the neighbouring frame).  Replaces temporal_sign + chan_partial + normalize_split.
//   gram   gram16y_kernel    sign(V V^T - T): 256 x 128 workgroup tiles (wave tiles 64 x 64: every LDS fragment feeds two
//                            MFMAs), two workgroups per CU, operands by LDS-DMA into 3-slot rings of 24 KB slots, swizzled
//                            32-byte rows (no pad bytes in the stream), upper triangle + mirrored tile; the epilogue turns
//                            G - T into sign bytes and stores both positions from registers (mirror pieces after a
//                            v_permlane32_swap, direct pieces after a transposition on the matrix pipe: no LDS, no barrier).
//                            gram16z_kernel: the same on 128 x 128 tiles / 4 waves / three workgroups per CU for launches
//                            that do not fill the chip twice.  gram16s_kernel for planes <= 256 pixels (64 x 64 tiles,
//                            wave-level split K, operands straight from L2 into registers).
//   sv     sv16b / sv16      dV^T = 2c V^T S, and in the epilogue the partial sums of <V, dV> per pixel over the workgroup's
//                            128 channels (the norm backward needs the full sum: one more pass over x and dV before), its V
//                            operands copied through the free LDS ring.
//   adam   opt_adam_kernel   temporal gradient from the signs + CSR rows, norm backward, Adam, and the partial sums of
//                            squares of the UPDATED features for the next prep (m / v / dV non-temporal on the big planes).
// prep and adam blocks share the loads of the per-pixel partial sums among their four slices (LDS exchange).
//
// All reductions have a fixed order: results are bit-reproducible run to run.
#include "opt_shared.h"
#include <stdlib.h>

namespace fresco {

constexpr int FAST_MAX_PART = 32;  // slices of the workspace's `part` / `dotp` arrays

bool opt_fast_ok(int C, int h, int w, int has_s) {
    static const int off = [] {
        const char* e = getenv("FRESCO_OPT_GENERIC");
        return (e && e[0] == '1') ? 1 : 0;
    }();
    const int hw = h * w;
    return !off && has_s && hw % 64 == 0 && C % 8 == 0 && 2 * ((C + 127) / 128) <= FAST_MAX_PART;
}

// Work split of prep / adam: a thread owns K channel octets of one pixel, a 256-thread block 64 pixels x 4 such slices.
// NPART = slices per pixel, NPB = blocks per pixel.  The ADAM split also fixes the partial sums of squares (one per Adam
// block, <= FAST_MAX_PART per pixel) that the next prep adds up, i.e. it is part of the arithmetic: it follows the WHOLE
// batch (Bg planes), not the launch's share of it, so that a CFG half on its own stream, or a rank's frame shard, rounds
// exactly as the undivided batch does.  K: few octets on planes too small to fill the chip otherwise (8 x 8, 16 x 16:
// the launch is a latency chain of K dependent octet rounds); 5 on the big planes (taps / CSR rows are set up once per
// thread).  Measured: adam alone at K = 3 / 4 / 5 / 8 / 10: 282 / 280 / 269 / 286 / 261 us at (640, 64^2), 164 / 156 / 149 /
// 164 / 139 at (1280, 32^2), prep flat between 4 and 10 -- but with the two CFG halves on two streams (the shipping
// form) K = 10 for adam makes the layer SLOWER (29.2 -> 29.5 ms: fewer, longer blocks beside the other half's
// launches), so both keep 5.
static void fast_slices(int C, int Bg, int hw, int cap, int* K, int* NPART, int* NPB) {
    const int C8 = C / 8;
    int64_t k = (int64_t)Bg * hw * C8 / 262144;
    k = k < 1 ? 1 : (k > cap ? cap : k);
    while (((C8 + k - 1) / k + 3) / 4 > FAST_MAX_PART) ++k;
    *K = (int)k;
    *NPART = (int)((C8 + k - 1) / k);
    *NPB = (*NPART + 3) / 4;
}
constexpr int PREP_K = 5, ADAM_K = 5;

// sum over the 4 slices of a block, in slice order; valid in the threads of slice 0
__device__ __forceinline__ float slice_sum_4(float v, int px, int sl, float (*red)[64]) {
    red[sl][px] = v;
    __syncthreads();
    return red[0][px] + red[1][px] + red[2][px] + red[3][px];
}

// ------------------------------------------------------------------------------------------------
// part[b][blockIdx.y][p] = sum of x^2 over the block's 4 channel slices (first iteration only).  grid (hw/64, NPB, B)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ cs, float* __restrict__ part, int C,
                                                             int hw, int K, int NPART) {
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + px, j = blockIdx.y * 4 + sl, b = blockIdx.z;
    __shared__ float red[4][64];
    const int cb = min(j * 8 * K, C), ce = min(cb + 8 * K, C);
    float acc = 0.f;
    for (int c = cb; c < ce; ++c) {
        const float x = cs[((int64_t)b * C + c) * hw + p];
        acc = fmaf(x, x, acc);
    }
    const float tot = slice_sum_4(acc, px, sl, red);
    if (sl == 0) part[((int64_t)b * gridDim.y + blockIdx.y) * hw + p] = tot;
}

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  prep / adam blocks own 64 consecutive pixels
// of a plane and gather from the rows just above / below (bilinear taps, CSR sources): give every XCD a CONTIGUOUS range
// of pixel blocks so that neighbouring rows are fetched into ONE L2 instead of three.
__device__ __forceinline__ int xcd_contiguous(int bx, int nx) { return nx % 8 == 0 ? (bx % 8) * (nx / 8) + bx / 8 : bx; }

// ------------------------------------------------------------------------------------------------
// prep: grid (hw/64, ceil(NPART/4), nck * (has_t ? n_pairs : n_loc)), 256 threads = 64 pixels x 4 channel slices.
// The thread of (pixel p, slice j, pair pj) owns channels [8 K j, 8 K (j+1)) of pixel p of the pair's FIRST frame: it
// normalises them (when that frame is local) and evaluates both residual signs of the pair for them.
// ------------------------------------------------------------------------------------------------
struct PrepArgs {
    const float* cs;
    const float* part;
    float* nrm;
    half_t *vh, *vl, *vph, *vpl;
    const float *bwd_flow, *fwd_flow, *bwd_occ, *fwd_occ;
    int8_t *sgn1, *sgn2;
    float* loss;
    TLayout L;
    int C, h, w, K, NPART, NPB, has_t, pm_tiled, cm_tiled;
    // frame-sharded form only: 0 = everything; 1 = the part that needs no halo frame (normalisation + copies of every
    // local frame, signs of the interior pairs); 2 = the signs of the two pairs that touch a halo frame, nothing else.
    // 1 followed by 2 performs, per element, exactly the operations of 0 (results identical bit for bit): the split only
    // lets the neighbour exchange of the halo frames run under the Gram / S V launches (fresco_opt_sharded_step_part)
    int phase;
};

__global__ __launch_bounds__(256) void opt_prep_kernel(PrepArgs a) {
    const int hw = a.h * a.w, C = a.C, C8 = C >> 3;
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    // Which blocks share an XCD's L2 at the same time: workgroup ids go round-robin to the 8 XCDs (id % 8); within an XCD
    // the PAIR index runs fastest, then the XCD's contiguous range of pixel blocks, then the channel groups -- the two
    // frames of pair j are the frames of pairs j - 1 and j + 1, and the taps reach into the rows of the neighbouring
    // pixel blocks (grid order x, y, z with XCD-contiguous x: 169.7 -> 166.5 us at (640, 64^2), 95 -> 92.5 at
    // (1280, 32^2); channel groups before pixel blocks: 187)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (gridDim.x % 8 == 0) {
        const int nx = gridDim.x, nz = gridDim.z;
        const int id = blockIdx.x + nx * (blockIdx.y + gridDim.y * blockIdx.z);  // dispatch order
        const int k = id & 7, s = id >> 3, rpx = nx >> 3;
        bz = s % nz;
        by = s / (nz * rpx);
        bx = k * rpx + (s / nz) % rpx;
    }
    const int p = bx * 64 + px;
    const int j = by * 4 + sl;
    const TLayout& L = a.L;
    const int NPZ = a.has_t ? L.n_pairs : L.n_loc;
    const int ck = bz / NPZ, pj = bz % NPZ;
    const int fn = (L.circular || !a.has_t) ? pj : pj - 1;  // the local frame this thread normalises (-1: a halo frame)
    const int sa = pj, sb = L.circular ? (pj + 1) % L.n_loc : pj + 1;
    float lsum = 0.f;
    __shared__ __attribute__((aligned(16))) half_t cmx[4][2][8][64];  // [slice = wave][hi, lo][channel of the octet][pixel]
    __shared__ float nred[4][64];
    // (all block-uniform)
    const bool boundary = a.has_t && !L.circular && (pj == 0 || pj == L.n_pairs - 1);  // the pair reads a halo frame
    const bool do_signs = a.has_t && !(a.phase == 1 && boundary) && !(a.phase == 2 && !boundary);
    const bool do_norm = fn >= 0 && a.phase != 2;
    if (!do_signs && !do_norm) return;
    const int64_t bn = (int64_t)ck * L.n_loc + fn;
    float n = 1.f;
    if (do_norm) {
        // |x[p]|^2 = the NPB partial sums the previous adam launch left behind.  The four slices of the block share the
        // loads (slice sl adds partials sl, sl + 4, ...) and exchange their sums through LDS: a thread that loads all NPB
        // (20 at C = 1280) itself spends 5 - 10 us of the small-plane launches on them (profiles/r04_opt_ablation.txt).
        // Every block adds in the same order.
        float ss = 0.f;
        for (int s = sl; s < a.NPB; s += 4) ss += a.part[(bn * a.NPB + s) * hw + p];
        nred[sl][px] = ss;
        __syncthreads();
        n = sqrtf((nred[0][px] + nred[1][px]) + (nred[2][px] + nred[3][px]));
    }
    if (j < a.NPART) {
        if (do_norm && j == 0) a.nrm[bn * hw + p] = n;
        OTaps tb, tf;
        float mb = 0.f, mf = 0.f;
        if (do_signs) {
            const float* fb = a.bwd_flow + (int64_t)pj * 2 * hw;
            const float* ff = a.fwd_flow + (int64_t)pj * 2 * hw;
            tb = otaps(fb[p], fb[hw + p], p % a.w, p / a.w, a.h, a.w);
            tf = otaps(ff[p], ff[hw + p], p % a.w, p / a.w, a.h, a.w);
            mb = 1.f - a.bwd_occ[(int64_t)pj * hw + p];
            mf = 1.f - a.fwd_occ[(int64_t)pj * hw + p];
        }
        const int o_end = min((j + 1) * a.K, C8);
        for (int o = j * a.K; o < o_end; ++o) {
            const int c0 = o * 8;
            const float* c1p = frame_plane(a.cs, L, ck, sa, c0, C, hw);
            float x1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x1[k] = c1p[(int64_t)k * hw + p];
            if (do_signs) {
                const float* c2p = frame_plane(a.cs, L, ck, sb, c0, C, hw);
                uint64_t w1 = 0, w2 = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float* q1 = c1p + (int64_t)k * hw;
                    const float* q2 = c2p + (int64_t)k * hw;
                    const float r1 = (q2[p] - osample(q1, tb)) * mb;
                    const float r2 = (x1[k] - osample(q2, tf)) * mf;
                    w1 |= (uint64_t)(uint8_t)(int8_t)sgn(r1) << (8 * k);
                    w2 |= (uint64_t)(uint8_t)(int8_t)sgn(r2) << (8 * k);
                    lsum += fabsf(r1) + fabsf(r2);
                }
                const int64_t so = ((int64_t)bz * C8 + o) * hw + p;  // signs: [pair][C/8][hw][8] bytes
                reinterpret_cast<uint64_t*>(a.sgn1)[so] = w1;
                reinterpret_cast<uint64_t*>(a.sgn2)[so] = w2;
            }
            if (do_norm) {
                half8_t h8, l8;
                // channel-major copies: the wave (= one slice: 64 consecutive pixels x 8 channels) transposes its octet
                // through its own LDS rows so that a lane stores 8 consecutive PIXELS of one channel -- one 16-byte unit of
                // either layout -- instead of 16 two-byte stores per thread
                half_t* th = &cmx[sl][0][0][0];
                half_t* tl = &cmx[sl][1][0][0];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float val = x1[k] / n;
                    const half_t hi16 = (half_t)val;
                    const half_t lo16 = (half_t)(val - (float)hi16);
                    h8[k] = hi16;
                    l8[k] = lo16;
                    th[k * 64 + px] = hi16;
                    tl[k * 64 + px] = lo16;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
                    const int kk = px >> 3, g = px & 7, c = c0 + kk, P = p - px + g * 8;
                    const half8_t uh = *reinterpret_cast<const half8_t*>(th + kk * 64 + g * 8);
                    const half8_t ul = *reinterpret_cast<const half8_t*>(tl + kk * 64 + g * 8);
                    const int64_t ov = a.cm_tiled ? ((((int64_t)bn * (C / 128) + c / 128) * (hw / 32) + P / 32) * 128 + c % 128) * 32 +
                                                        (((((P & 31) >> 3) ^ (c >> 2)) & 3) << 3)
                                                  : ((int64_t)bn * C + c) * hw + P;
                    *reinterpret_cast<half8_t*>(a.vh + ov) = uh;
                    *reinterpret_cast<half8_t*>(a.vl + ov) = ul;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // (the rows are rewritten by the next octet)
                // pixel-major: the octet is one 16-byte unit of the pixel's row
                const int64_t op = a.pm_tiled ? ((((int64_t)bn * (hw / 128) + p / 128) * (C / 16) + c0 / 16) * 128 + p % 128) * 16 +
                                                    ((((c0 % 16) >> 3) ^ ((p >> 3) & 1)) << 3)
                                              : ((int64_t)bn * hw + p) * C + c0;
                *reinterpret_cast<half8_t*>(a.vph + op) = h8;
                *reinterpret_cast<half8_t*>(a.vpl + op) = l8;
            }
        }
    }
    if (a.loss) {
        __shared__ float red[4];
        const float tot = block_sum_256(lsum, red);
        if (threadIdx.x == 0) atomicAdd(a.loss, tot);
    }
}

// ------------------------------------------------------------------------------------------------
// adam: grid (hw/64, ceil(NPART/4), planes), 256 threads = 64 pixels x 4 channel slices.
//   g = grad_t (if has_t) + (dV - V <V,dV>)/|X| (if has_s);  <V,dV>[b][p] = sum of the NCT partials of the S V epilogue
// mode 0: Adam update of cs, m, v + partial sum of squares of the new cs;  mode 1: write g to gout (loss_grad entry)
// ------------------------------------------------------------------------------------------------
struct AdamKArgs {
    float *cs, *m, *v2;
    TGradArgs tg;
    const float *dvt, *nrm, *dotp;
    float *part, *gout;
    int C, hw, K, NPART, NCT, has_t, has_s, mode;
    AdamArgs a;
};

// NT: m, v and dV -- read once and (m, v) written once per iteration -- carry the non-temporal hint on planes too big for
// the caches (>= 64 MB of features: 32 x 32 and 64 x 64): they stop evicting x and the sign words, which the same
// launch and the next prep re-read, and this instantiation fits 128 registers = four waves per SIMD (adam 295 -> 254 us
// and the following prep 186 -> 180 at (640, 64^2), 160 -> 145 / 92 -> 84 at (1280, 32^2)); on the 8 x 8 / 16 x 16 planes
// everything is cache-resident and the hint costs 3 us per launch.  The same hint on prep's copy / sign stores: slower
// everywhere (163 -> 173, 36 -> 64 at 16 x 16); on the dV stores of the S V kernel: -8 us of the Adam launch at 32 x 32, +4
// at 16 x 16, nothing at 64 x 64: neither kept.
template <bool NT>
__global__ __launch_bounds__(256) void opt_adam_kernel(AdamKArgs k) {
    const int hw = k.hw, C = k.C, C8 = C >> 3;
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int p = xcd_contiguous(blockIdx.x, gridDim.x) * 64 + px, j = blockIdx.y * 4 + sl, b = blockIdx.z;
    __shared__ float red[4][64], dred[4][64];
    float dot = 0.f, inv_n = 0.f, n = 1.f;
    if (k.has_s) {  // <V, dV>[p] = the NCT partials of the S V epilogue: the four slices share the loads (as prep does)
        float d = 0.f;
        for (int s = sl; s < k.NCT; s += 4) d += k.dotp[((int64_t)b * k.NCT + s) * hw + p];
        dred[sl][px] = d;
        n = k.nrm[(int64_t)b * hw + p];
        inv_n = 1.f / n;
        __syncthreads();
        dot = (dred[0][px] + dred[1][px]) + (dred[2][px] + dred[3][px]);
    }
    TGradPixel tp;
    if (k.has_t) tp.init(k.tg, b, p, hw);
    const AdamArgs a = k.a;
    float ss = 0.f;
    const int o_end = min((j + 1) * k.K, C8);
    for (int o = min(j * k.K, C8); o < o_end; ++o) {
        float tgv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k.has_t) tp.values(k.tg, o, p, C8, hw, tgv);
        const int64_t o0 = ((int64_t)b * C + o * 8) * hw + p;
        float x[8], dv[8], mo[8], vo[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            x[q] = k.cs[o0 + (int64_t)q * hw];
            const int64_t oq = o0 + (int64_t)q * hw;
            dv[q] = !k.has_s ? 0.f : (NT ? __builtin_nontemporal_load(k.dvt + oq) : k.dvt[oq]);
            if (k.mode == 0) {
                mo[q] = NT ? __builtin_nontemporal_load(k.m + oq) : k.m[oq];
                vo[q] = NT ? __builtin_nontemporal_load(k.v2 + oq) : k.v2[oq];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float g = tgv[q];
            if (k.has_s) g += (dv[q] - (x[q] / n) * dot) * inv_n;  // V = X/|X| rebuilt: the quotient prep rounded
            if (k.mode == 1) {
                k.gout[o0 + (int64_t)q * hw] = g;
            } else {
                const float mm = a.beta1 * mo[q] + (1.f - a.beta1) * g;
                const float vv = a.beta2 * vo[q] + (1.f - a.beta2) * g * g;
                if (NT) {
                    __builtin_nontemporal_store(mm, k.m + o0 + (int64_t)q * hw);
                    __builtin_nontemporal_store(vv, k.v2 + o0 + (int64_t)q * hw);
                } else {
                    k.m[o0 + (int64_t)q * hw] = mm;
                    k.v2[o0 + (int64_t)q * hw] = vv;
                }
                const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
                const float xn = x[q] - a.step_size * (mm / denom);
                k.cs[o0 + (int64_t)q * hw] = xn;
                ss = fmaf(xn, xn, ss);
            }
        }
    }
    if (k.mode == 0 && k.has_s) {  // (uniform: every thread of the block gets here)
        const float tot = slice_sum_4(ss, px, sl, red);
        if (sl == 0) k.part[((int64_t)b * gridDim.y + blockIdx.y) * hw + p] = tot;
    }
}

// Where the 16-byte piece S[p][q .. q + 15] (q % 16 == 0) of the sign matrix goes.  Plain layout: row-major.  Tiled (what
// sv16b_kernel streams): [plane][pixel tile of 256][chunk of 32 q][256 rows p][32 bytes], the four 8-byte units of a row
// XOR-swizzled with (p >> 3) & 3: the piece lands in 16-byte half ((q >> 4) & 1) ^ (g >> 1) and its two units are swapped if g & 1.
__device__ __forceinline__ void s_store_piece(int8_t* __restrict__ sgn_out, u32x4 v, int b, int p, int q, int hw, int s_tiled) {
    int64_t so;
    if (s_tiled) {
        const int g = (p >> 3) & 3;
        so = ((((int64_t)b * (hw / 256) + p / 256) * (hw / 32) + q / 32) * 256 + p % 256) * 32 + ((((q >> 4) & 1) ^ (g >> 1)) << 4);
        if (g & 1) v = u32x4{v[2], v[3], v[0], v[1]};
    } else {
        so = ((int64_t)b * hw + p) * hw + q;
    }
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(sgn_out + so));
}

// ------------------------------------------------------------------------------------------------
// Gram step for planes of <= 256 pixels (the 8 x 8 and 16 x 16 inputs of up_blocks.0 / 1): few tiles, long K
// (C = 1280), so the launch is a latency chain, not a throughput problem.  64 x 64 tiles, ALL nt x nt of them (no
// mirroring); the NW waves of a workgroup split K (wave w takes the k16 steps w, w + NW, ...), each computing the whole
// tile with operands loaded straight from L2 into MFMA fragments (plain pixel-major layout, 16 bytes per lane, next
// step's loads in flight during the 12 MFMAs of the current one); the NW partial tiles are summed in wave order through
// LDS.  grid (nt*nt, 1, B), NW*64 threads, dynamic LDS NW*64*68*4 bytes.
// ------------------------------------------------------------------------------------------------
constexpr int GS_RS = 68;  // floats per LDS row of a partial tile (64 + 4: conflict-free 16-byte reads)

template <int NW>
__global__ __launch_bounds__(NW * 64) void gram16s_kernel(const half_t* __restrict__ vph, const half_t* __restrict__ vpl,
                                                          const float* __restrict__ target, int8_t* __restrict__ sgn_out,
                                                          float* __restrict__ loss, int C, int hw, int s_tiled) {
    extern __shared__ __attribute__((aligned(16))) float gs_red[];  // [NW][64][GS_RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = hw / 64;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt, b = blockIdx.z;
    const int p0 = ti * 64, q0 = tj * 64;
    const half_t* hb = vph + (int64_t)b * hw * C;
    const half_t* lb = vpl + (int64_t)b * hw * C;
    const int64_t ra0 = (int64_t)(p0 + l31) * C + hi * 8, ra1 = ra0 + (int64_t)32 * C;
    const int64_t rb0 = (int64_t)(q0 + l31) * C + hi * 8, rb1 = rb0 + (int64_t)32 * C;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    struct Frag {
        half8_t ah[2], al[2], bh[2], bl[2];
    };
    auto loadf = [&](int ks, Frag& f) __attribute__((always_inline)) {
        const int k = ks * 16;
        f.ah[0] = *reinterpret_cast<const half8_t*>(hb + ra0 + k);
        f.ah[1] = *reinterpret_cast<const half8_t*>(hb + ra1 + k);
        f.al[0] = *reinterpret_cast<const half8_t*>(lb + ra0 + k);
        f.al[1] = *reinterpret_cast<const half8_t*>(lb + ra1 + k);
        f.bh[0] = *reinterpret_cast<const half8_t*>(hb + rb0 + k);
        f.bh[1] = *reinterpret_cast<const half8_t*>(hb + rb1 + k);
        f.bl[0] = *reinterpret_cast<const half8_t*>(lb + rb0 + k);
        f.bl[1] = *reinterpret_cast<const half8_t*>(lb + rb1 + k);
    };
    auto compute = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[jj], acc[i][jj], 0, 0, 0);
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[jj], acc[i][jj], 0, 0, 0);
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[jj], acc[i][jj], 0, 0, 0);
            }
    };
    const int nks = C / 16;
    Frag f0, f1;
    int ks = wave;
    if (ks < nks) loadf(ks, f0);
    while (ks < nks) {
        if (ks + NW < nks) loadf(ks + NW, f1);
        compute(f0);
        ks += NW;
        if (ks >= nks) break;
        if (ks + NW < nks) loadf(ks + NW, f0);
        compute(f1);
        ks += NW;
    }

    float* mine = gs_red + (size_t)wave * 64 * GS_RS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * GS_RS + jj * 32 + l31] = acc[i][jj][r];
    __syncthreads();
    float lsum = 0.f;
    if (tid < 256) {  // 256 pieces of 16 consecutive entries of one row
        const int row = tid >> 2, cq = (tid & 3) * 16;
        float g[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
        for (int w = 0; w < NW; ++w) {  // fixed order
            const float* src = gs_red + ((size_t)w * 64 + row) * GS_RS + cq;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const floatx4 t = *reinterpret_cast<const floatx4*>(src + e4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e4 * 4 + e] += t[e];
            }
        }
        const int gp = p0 + row, gq = q0 + cq;
        const float* tg = target + ((int64_t)b * hw + gp) * hw + gq;
        u32x4 packed;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(tg + e4 * 4);
            uint32_t wv = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = g[e4 * 4 + e] - t[e];
                lsum += fabsf(d);
                wv |= (uint32_t)(uint8_t)sign_byte(d) << (8 * e);
            }
            packed[e4] = wv;
        }
        s_store_piece(sgn_out, packed, b, gp, gq, hw, s_tiled);
    }
    if (loss) {
        const float tot = wave_sum(lsum);
        __syncthreads();
        if (lane == 0) gs_red[wave] = tot;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += gs_red[w];
            atomicAdd(loss, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gram16s_kernel with COALESCED operand loads (round 6).  gram16s loads its MFMA fragments straight from global memory, a
// lane per row: 64 rows one row stride apart per instruction, which a CU's texture-address path serves at ~9 B/clk
// (profiles/r06_kvproj_ablation.txt; 256 workgroups x 640 KB in 30 us at (1280, 16^2) is exactly that rate).  Here the
// workgroup loads the NW k-steps of a round -- one per wave: [rows][NW * 16 channels] = NW * 32 contiguous bytes per row --
// cooperatively and coalesced (consecutive lanes = consecutive 16 bytes of a row) into LDS, double-buffered through
// registers, and every wave reads ITS k-step's fragments from there: the same k-steps, the same three products per step in
// the same order, the same reduction -- bit-identical signs (tested against gram16s_kernel).  LDS: two buffers of
// [A hi | A lo | B hi | B lo][64 rows][NW * 32 + 16 bytes] (odd 16-byte row stride), later the partial-tile buffer.
// grid (nt*nt, 1, B), NW*64 threads.
// ------------------------------------------------------------------------------------------------
template <int NW>
struct GramCCfg {
    static constexpr int RS = NW * 32 + 16;          // LDS bytes per staged row
    static constexpr int BLK = 64 * RS;              // one operand block
    static constexpr int BUF = 4 * BLK;
    static constexpr int PPR = NW * 2;               // 16-byte pieces per row and round
    static constexpr int NLD = 4 * 64 * PPR / (NW * 64);  // loads per thread and round (= 8)
    static constexpr int RED = NW * 64 * GS_RS * 4;  // the partial-tile buffer of the reduction
    static constexpr int LDS = 2 * BUF > RED ? 2 * BUF : RED;
};

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void gram16c_kernel(const half_t* __restrict__ vph, const half_t* __restrict__ vpl,
                                                          const float* __restrict__ target, int8_t* __restrict__ sgn_out,
                                                          float* __restrict__ loss, int C, int hw, int s_tiled) {
    using G = GramCCfg<NW>;
    extern __shared__ __attribute__((aligned(16))) char gc_smem[];
    float* gs_red = reinterpret_cast<float*>(gc_smem);  // (after the K loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = hw / 64;
    // Workgroup ids go round-robin to the 8 XCDs, each with its own L2: with the tile index fastest, the nt * nt tiles of a
    // plane land on all eight, and every L2 pulls every plane's operand rows through the fabric (8 x 21 MB at (1280, 16^2):
    // the launch ran at the fabric's rate, not the CUs').  All tiles of a plane on ONE XCD when the planes divide by 8.
    int tile = blockIdx.x, b = blockIdx.z;
    if (gridDim.z % 8 == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.z;
        const int xcd = lin & 7, slot = lin >> 3;
        b = xcd + 8 * (slot / (int)gridDim.x);
        tile = slot % (int)gridDim.x;
    }
    const int ti = tile / nt, tj = tile % nt;
    const int p0 = ti * 64, q0 = tj * 64;
    const half_t* hb = vph + (int64_t)b * hw * C;
    const half_t* lb = vpl + (int64_t)b * hw * C;
    typedef unsigned int gu4_t __attribute__((ext_vector_type(4)));

    // this thread's NLD pieces of a round: (block, row, piece) -> source element offset (without the round's k offset),
    // LDS byte offset inside a buffer
    int64_t soff[G::NLD];
    int doff[G::NLD], kpc[G::NLD];
    const half_t* sbase[G::NLD];
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
        const int q = j * (NW * 64) + tid;
        const int blk = q / (64 * G::PPR), rem = q % (64 * G::PPR);
        const int row = rem / G::PPR, pc = rem % G::PPR;
        sbase[j] = (blk & 1) ? lb : hb;                                   // blocks: A hi, A lo, B hi, B lo
        soff[j] = (int64_t)((blk < 2 ? p0 : q0) + row) * C + pc * 8;
        kpc[j] = pc * 8;
        doff[j] = blk * G::BLK + row * G::RS + pc * 16;
    }
    const int nks = C / 16;
    const int rounds = (nks + NW - 1) / NW;
    // Three rounds of loads are in flight (three register sets), two rounds live in LDS: a round's products are ~0.2 us of
    // matrix work, an L2 round trip ~2 us -- one round ahead left the loop latency-bound (30 -> 25 us at (1280, 16^2))
#define GC_LOAD(RR_, TT_)                                                                          \
    {                                                                                              \
        const int k0_ = (RR_) * NW * 16;                                                           \
        _Pragma("unroll") for (int j = 0; j < G::NLD; ++j) {                                        \
            gu4_t v_ = {0u, 0u, 0u, 0u};                                                           \
            if ((RR_) < rounds && k0_ + kpc[j] < C)                                                \
                v_ = *reinterpret_cast<const gu4_t*>(sbase[j] + soff[j] + k0_);                    \
            TT_[j] = v_;                                                                           \
        }                                                                                          \
    }
#define GC_STORE(BI_, TT_)                                                                         \
    _Pragma("unroll") for (int j = 0; j < G::NLD; ++j)                                              \
        *reinterpret_cast<gu4_t*>(gc_smem + (BI_) * G::BUF + doff[j]) = TT_[j];
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    auto compute = [&](int r) __attribute__((always_inline)) {
        const int ks = r * NW + wave;
        if (ks < nks) {
            const char* L = gc_smem + (r & 1) * G::BUF + wave * 32 + hi * 16;
            half8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8_t*>(L + 0 * G::BLK + (i * 32 + l31) * G::RS);
                al[i] = *reinterpret_cast<const half8_t*>(L + 1 * G::BLK + (i * 32 + l31) * G::RS);
                bh[i] = *reinterpret_cast<const half8_t*>(L + 2 * G::BLK + (i * 32 + l31) * G::RS);
                bl[i] = *reinterpret_cast<const half8_t*>(L + 3 * G::BLK + (i * 32 + l31) * G::RS);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {  // (the product order of gram16s_kernel)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jj], acc[i][jj], 0, 0, 0);
                }
        }
    };

    gu4_t s0[G::NLD], s1[G::NLD], s2[G::NLD];
    GC_LOAD(0, s0)
    GC_LOAD(1, s1)
    GC_LOAD(2, s2)
    __builtin_amdgcn_sched_barrier(0);  // (hipcc sinks loads to their uses: keep the batches in front)
    GC_STORE(0, s0)
    __syncthreads();
    // round r reads LDS buffer r & 1; the set that held round r + 1 is written to the other buffer behind the products of
    // round r (its readers finished in round r - 1: the barrier at the end of that round ordered them)
    for (int r = 0; r < rounds; r += 3) {
        GC_LOAD(r + 3, s0)
        __builtin_amdgcn_sched_barrier(0);
        compute(r);
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 >= rounds) break;
        GC_STORE((r + 1) & 1, s1)
        __syncthreads();
        GC_LOAD(r + 4, s1)
        __builtin_amdgcn_sched_barrier(0);
        compute(r + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (r + 2 >= rounds) break;
        GC_STORE((r + 2) & 1, s2)
        __syncthreads();
        GC_LOAD(r + 5, s2)
        __builtin_amdgcn_sched_barrier(0);
        compute(r + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (r + 3 >= rounds) break;
        GC_STORE((r + 3) & 1, s0)
        __syncthreads();
    }
#undef GC_LOAD
#undef GC_STORE
    __syncthreads();  // every wave is done with the staging buffers: they become the partial-tile buffer

    // ---- reduction, target, signs: gram16s_kernel's epilogue (the staging buffers are dead: the loop ended on a barrier)
    float* mine = gs_red + (size_t)wave * 64 * GS_RS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * GS_RS + jj * 32 + l31] = acc[i][jj][r];
    __syncthreads();
    float lsum = 0.f;
    if (tid < 256) {  // 256 pieces of 16 consecutive entries of one row
        const int row = tid >> 2, cq = (tid & 3) * 16;
        float g[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
        for (int w = 0; w < NW; ++w) {  // fixed order
            const float* src = gs_red + ((size_t)w * 64 + row) * GS_RS + cq;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const floatx4 t = *reinterpret_cast<const floatx4*>(src + e4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e4 * 4 + e] += t[e];
            }
        }
        const int gp = p0 + row, gq = q0 + cq;
        const float* tg = target + ((int64_t)b * hw + gp) * hw + gq;
        u32x4 packed;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(tg + e4 * 4);
            uint32_t wv = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = g[e4 * 4 + e] - t[e];
                lsum += fabsf(d);
                wv |= (uint32_t)(uint8_t)sign_byte(d) << (8 * e);
            }
            packed[e4] = wv;
        }
        s_store_piece(sgn_out, packed, b, gp, gq, hw, s_tiled);
    }
    if (loss) {
        const float tot = wave_sum(lsum);
        __syncthreads();
        if (lane == 0) gs_red[wave] = tot;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += gs_red[w];
            atomicAdd(loss, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same Gram step for planes of ONE 64 x 64 tile (the 8 x 8 input of up_blocks.0) as two launches (round 6).  At
// batch 16 gram16s_kernel is 16 workgroups: 16 CUs issue all of the launch's row-strided 16-byte fragment loads, which a
// CU's texture-address unit sustains at ~9 B/clk (profiles/r06_kvproj_ablation.txt measured the same rate for the same
// access pattern) -- 26 us for 31 Mflop per plane while 240 CUs idle.  Here every split-K slice is a one-wave workgroup
// of its own (grid (tiles, NW, planes): 128 CUs share the loads), leaving its partial tile in the workspace, and a second
// launch adds the NW partials in wave order and writes the signs.  Each partial is the sum gram16s_kernel<NW>'s wave w
// forms -- same k-steps, same MFMA order -- and they are added in the same order: the signs are bit-identical (tested).
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64) void gram16sp_kernel(const half_t* __restrict__ vph, const half_t* __restrict__ vpl,
                                                      float* __restrict__ gpart, int C, int hw) {
    const int lane = threadIdx.x & 63, wave = blockIdx.y;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = hw / 64;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt, b = blockIdx.z;
    const int p0 = ti * 64, q0 = tj * 64;
    const half_t* hb = vph + (int64_t)b * hw * C;
    const half_t* lb = vpl + (int64_t)b * hw * C;
    const int64_t ra0 = (int64_t)(p0 + l31) * C + hi * 8, ra1 = ra0 + (int64_t)32 * C;
    const int64_t rb0 = (int64_t)(q0 + l31) * C + hi * 8, rb1 = rb0 + (int64_t)32 * C;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    struct Frag {
        half8_t ah[2], al[2], bh[2], bl[2];
    };
    auto loadf = [&](int ks, Frag& f) __attribute__((always_inline)) {
        const int k = ks * 16;
        f.ah[0] = *reinterpret_cast<const half8_t*>(hb + ra0 + k);
        f.ah[1] = *reinterpret_cast<const half8_t*>(hb + ra1 + k);
        f.al[0] = *reinterpret_cast<const half8_t*>(lb + ra0 + k);
        f.al[1] = *reinterpret_cast<const half8_t*>(lb + ra1 + k);
        f.bh[0] = *reinterpret_cast<const half8_t*>(hb + rb0 + k);
        f.bh[1] = *reinterpret_cast<const half8_t*>(hb + rb1 + k);
        f.bl[0] = *reinterpret_cast<const half8_t*>(lb + rb0 + k);
        f.bl[1] = *reinterpret_cast<const half8_t*>(lb + rb1 + k);
    };
    auto compute = [&](const Frag& f) __attribute__((always_inline)) {  // (the product order of gram16s_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[jj], acc[i][jj], 0, 0, 0);
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[jj], acc[i][jj], 0, 0, 0);
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[jj], acc[i][jj], 0, 0, 0);
            }
    };
    const int nks = C / 16;
    Frag f0, f1;
    int ks = wave;
    if (ks < nks) loadf(ks, f0);
    while (ks < nks) {
        if (ks + NW < nks) loadf(ks + NW, f1);
        compute(f0);
        ks += NW;
        if (ks >= nks) break;
        if (ks + NW < nks) loadf(ks + NW, f0);
        compute(f1);
        ks += NW;
    }
    float* mine = gpart + (((int64_t)b * gridDim.x + blockIdx.x) * NW + wave) * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + jj * 32 + l31] = acc[i][jj][r];
}

// grid (tiles, 1, planes), 256 threads: thread = 16 consecutive entries of one row (as gram16s_kernel's reduction)
__global__ __launch_bounds__(256) void gram16sr_kernel(const float* __restrict__ gpart, const float* __restrict__ target,
                                                       int8_t* __restrict__ sgn_out, int NW, int hw, int s_tiled) {
    const int tid = threadIdx.x;
    const int nt = hw / 64;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt, b = blockIdx.z;
    const int p0 = ti * 64, q0 = tj * 64;
    const int row = tid >> 2, cq = (tid & 3) * 16;
    const float* part = gpart + ((int64_t)b * gridDim.x + blockIdx.x) * NW * 4096;
    float g[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) g[e] = 0.f;
    for (int w = 0; w < NW; ++w) {  // fixed order
        const float* src = part + (int64_t)w * 4096 + row * 64 + cq;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(src + e4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e4 * 4 + e] += t[e];
        }
    }
    const int gp = p0 + row, gq = q0 + cq;
    const float* tg = target + ((int64_t)b * hw + gp) * hw + gq;
    u32x4 packed;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const floatx4 t = *reinterpret_cast<const floatx4*>(tg + e4 * 4);
        uint32_t wv = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) wv |= (uint32_t)(uint8_t)sign_byte(g[e4 * 4 + e] - t[e]) << (8 * e);
        packed[e4] = wv;
    }
    s_store_piece(sgn_out, packed, b, gp, gq, hw, s_tiled);
}

// ------------------------------------------------------------------------------------------------
// Gram step for the big planes (hw % 256 == 0, hw >= 512, C % 32 == 0): gram16y_kernel below.  Shared pieces: the tile
// walk -- tiles (ti, tj) of 256 x 128 pixels with tj >= 2 ti in units of 128 pixels; the 128-row half of a tile that lies
// BELOW the diagonal (only when tj == 2 ti) is the mirror image of its neighbour's upper half and is not written; halves
// above the diagonal are also written transposed to their mirror position; XCD-contiguous walk in 1024 x 1024 super-tiles.
// ------------------------------------------------------------------------------------------------

static int gx_tiles_per_plane(int hw) {
    const int n128 = hw / 128, n256 = hw / 256;
    return n256 * n128 - n256 * (n256 - 1);
}

__device__ __forceinline__ void gx_tile(int idx, int hw, int& ti, int& tj) {
    if (hw % 1024 == 0) {
        const int ns = hw / 1024;
        int si = 0;
        for (;; ++si) {  // super-row si: its diagonal block (20 tiles), then ns - 1 - si full blocks (4 x 8 tiles)
            const int row_tiles = 20 + 32 * (ns - 1 - si);
            if (idx < row_tiles) break;
            idx -= row_tiles;
        }
        if (idx < 20) {
            int r = 0;
            while (idx >= 8 - 2 * r) {
                idx -= 8 - 2 * r;
                ++r;
            }
            ti = si * 4 + r;
            tj = si * 8 + 2 * r + idx;
        } else {
            idx -= 20;
            const int sj = si + 1 + idx / 32;
            ti = si * 4 + (idx % 32) / 8;
            tj = sj * 8 + idx % 8;
        }
    } else {
        const int n128 = hw / 128;
        ti = 0;
        while (idx >= n128 - 2 * ti) {
            idx -= n128 - 2 * ti;
            ++ti;
        }
        tj = 2 * ti + idx;
    }
}

template <int N_>
__device__ __forceinline__ void gx_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N_) : "memory");
}

// ------------------------------------------------------------------------------------------------
// Epilogue of the 256 x 128 / 128 x 128 Gram kernels for one wave's 64 x 64 sub-tile (rows wm * 64 .., columns wn * 64 ..
// of the workgroup tile): G - T -> sign bytes, stored from registers in both positions -- no LDS, no barrier.
//   * the targets of 32 x 32 block n + 1 are requested before block n is turned into signs: the epilogue waits for
//     about one memory round trip instead of four -- while a workgroup waits there its neighbour has the CU alone and
//     cannot fill the matrix pipe (526 -> 512 us at (640, 64^2); two blocks ahead, or block 0 from inside the K loop,
//     spill and lose: 585 / 548 us; accumulators that START at -T, so that the epilogue has no target loads at all: 467
//     against 461 us, and more near-tie signs off -- the partial sums then live at |T| for the whole K loop;
//     profiles/r04_ab_opt_forms.txt);
//   * signs by arithmetic: s = med3(d * 2^126, -1, 1) (exactly -1, 0 or +1), cvt_pkrtz packs two of them as fp16, v_perm
//     keeps the high bytes (0x3C / 0xBC / 0x00: what the S V kernel expands);
//   * MIRROR position (wgt == 2; rows q, consecutive p): the accumulator layout gives a lane S[p .. p + 3][q] per dword,
//     for p = 8 r4 + 4 hi; one v_permlane32_swap per dword pair leaves lanes 0-31 with rows 0-15 and lanes 32-63 with rows
//     16-31 of the lane's column -- a 16-byte piece of mirror row q; a wave instruction writes 32 rows x 32 bytes = 1 KiB
//     contiguous of the tiled layout;
//   * DIRECT position (rows p, consecutive q): the 32 x 32 block is TRANSPOSED ON THE MATRIX PIPE.  The packed fp16
//     signs a lane holds are, as they are, the A operand of a 32 x 32 x 16 product whose row index is the lane's column
//     q and whose k slots are the lane's rows; against a constant 0 / 1 B operand that sends k slot (hi, j) of step t
//     to column 16 t + 8 (j >> 2) + 4 hi + (j & 3) -- the row that slot holds -- two MFMAs return S[p][q] with lanes
//     along p and registers along q: the same bytes, the same swap, one 16-byte store per lane.  2 MFMAs per 120 of the
//     K loop replace 16 one-byte LDS writes per lane and block, the staging tile, two workgroup barriers and the store
//     loop (a first form kept the LDS tile for this position: profiles/r04_gram_ablation.txt, "direct" 20 us of 455).
// Returns the lane's sum of |G - T| (LOSS).
// ------------------------------------------------------------------------------------------------
template <bool LOSS>
__device__ __forceinline__ float gram_sign_epilogue(const floatx16 (&acc)[2][2], const float* __restrict__ tgt, int hw, int wgt,
                                                     int wm, int wn, int l31, int hi, int8_t* __restrict__ sgn_out, int b,
                                                     int p0, int q0, int s_tiled) {
    float lsum = 0.f;
    if (wgt == 0) return lsum;  // (wave-uniform: a sub-tile below the diagonal is the mirror image of another tile's)
    half8_t bt[2];  // B operands of the transposing products
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) bt[t][j] = (16 * t + 8 * (j >> 2) + 4 * hi + (j & 3) == l31) ? (half_t)1.f : (half_t)0.f;
    float tnext[16];
    auto load_targets = [&](int blk) __attribute__((always_inline)) {
        const int i = blk >> 1, jj = blk & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tnext[r] = __builtin_nontemporal_load(tgt + (int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * hw + jj * 32);
    };
    auto bytes_of = [](float s0, float s1, float s2, float s3, uint32_t& h01, uint32_t& h23) __attribute__((always_inline)) {
        h01 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(s0, s1));
        h23 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(s2, s3));
        return __builtin_amdgcn_perm(h23, h01, 0x07050301u);  // the four high bytes
    };
    // dwords w[r4] = 4 consecutive entries at offset 8 r4 + 4 hi of a line of 32 -> the lane's 16 contiguous bytes (lanes
    // 0-31: entries 0-15, lanes 32-63: 16-31).  swap(a, b): lanes 0-31 get (own a, upper partner's a), lanes 32-63 (lower
    // partner's b, own b)
    auto piece_of = [](const uint32_t (&w)[4]) __attribute__((always_inline)) {
        const auto P = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
        const auto Q = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
        return u32x4{(uint32_t)P[0], (uint32_t)P[1], (uint32_t)Q[0], (uint32_t)Q[1]};
    };
    if (!LOSS) load_targets(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (LOSS) load_targets(i * 2 + jj);  // (the loss-reporting instantiation has no registers for the look-ahead)
            float tv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) tv[r] = tnext[r];
            if (!LOSS && i * 2 + jj < 3) load_targets(i * 2 + jj + 1);
            uint32_t sg[4], hp[8];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float s4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc[i][jj][r4 * 4 + e] - tv[r4 * 4 + e];
                    if (LOSS) lsum += fabsf(d);
                    s4[e] = __builtin_amdgcn_fmed3f(d * 0x1p126f, -1.f, 1.f);  // exactly -1, 0 or +1 for d = 0 and every normal d
                }
                sg[r4] = bytes_of(s4[0], s4[1], s4[2], s4[3], hp[2 * r4], hp[2 * r4 + 1]);
            }
            const int rowb = wm * 64 + i * 32, colb = wn * 64 + jj * 32;
            if (wgt == 2) s_store_piece(sgn_out, piece_of(sg), b, q0 + colb + l31, p0 + rowb + hi * 16, hw, s_tiled);
            floatx16 dt;
#pragma unroll
            for (int r = 0; r < 16; ++r) dt[r] = 0.f;
            dt = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, u32x4{hp[0], hp[1], hp[2], hp[3]}), bt[0], dt, 0, 0, 0);
            dt = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, u32x4{hp[4], hp[5], hp[6], hp[7]}), bt[1], dt, 0, 0, 0);
            uint32_t dg[4], u0, u1;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) dg[r4] = bytes_of(dt[r4 * 4], dt[r4 * 4 + 1], dt[r4 * 4 + 2], dt[r4 * 4 + 3], u0, u1);
            s_store_piece(sgn_out, piece_of(dg), b, p0 + rowb + l31, q0 + colb + hi * 16, hw, s_tiled);
        }
    return lsum;
}

// ------------------------------------------------------------------------------------------------
// gram16y_kernel: 256 x 128 workgroup tiles, 8 waves as 4 x 2 with wave tiles 64 x 64 (per k16 step a wave reads 4 + 4
// LDS fragments for 12 MFMAs; round 3's 128 x 128 / 64 x 32 form: 6 for 6), TWO workgroups per CU.  A first form with one
// workgroup per CU (K chunks of 32 in a 3 x 48 KB ring, targets prefetched into 64 registers; removed, see git history)
// measured 559 us at (640, 64^2) with the matrix pipe busy 44 %: its DMA waits, target stream and 6 us sign / store
// epilogue overlapped with nothing (ablations: profiles/r04_gram_ablation.txt, PMC of that removed kernel: profiles/r04_pmc_removed_gram16x_kernel.csv; the shipped kernels: profiles/r05_pmc_opt_*.csv).
// This form: 478 us when written, 447 - 462 us with the epilogue of gram_sign_epilogue (above).
//   * K chunks of 16 channels: a slot is six 4 KB blocks (24 KB), three slots = 72 KB per workgroup -> two per CU;
//     operands pre-tiled by prep as [plane][128-pixel tile][16-channel chunk][128][2 x 16 B], the two units of pixel row r
//     swapped when (r >> 3) & 1 (conflict-free ds_read_b128 on 32-byte rows);
//   * <= 128 registers: the target values are NOT prefetched into 64 registers; the epilogue requests them one 32 x 32
//     block ahead and the other workgroup's MFMAs cover the rest of the latency;
//   * sign bytes by arithmetic: med3(d * 2^64, -1, 1) -> cvt_pkrtz -> v_perm of the two high bytes (3.75 VALU per value
//     instead of ~10 compares / selects); target loads and sign stores carry the non-temporal hint (streams that
//     would otherwise evict the operand blocks the tiles of an XCD share in L2).
// grid (tiles per plane, 1, B), 512 threads, dynamic LDS 3 * 24 KB.
// ------------------------------------------------------------------------------------------------
constexpr int GY_BLK = 128 * 32;       // one (pixel tile, chunk) block of one array: 4 KB
constexpr int GY_SLOT = 6 * GY_BLK;    // Ah0 Ah1 Al0 Al1 Bh Bl
constexpr int GY_NS = 3;

template <bool LOSS>
__global__ __launch_bounds__(512, 4) void gram16y_kernel(const half_t* __restrict__ vph, const half_t* __restrict__ vpl,
                                                         const float* __restrict__ target, int8_t* __restrict__ sgn_out,
                                                         float* __restrict__ loss, int C, int hw, int s_tiled) {
    extern __shared__ __attribute__((aligned(16))) char gy_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int total = gridDim.x * gridDim.z;
    if (total % 8 == 0) lin = (lin % 8) * (total / 8) + lin / 8;  // XCD-contiguous ranges: the operand row blocks the tiles of an XCD share stay in ONE L2
    const int b = lin / gridDim.x;
    int ti, tj;
    gx_tile(lin % gridDim.x, hw, ti, tj);
    const int p0 = ti * 256, q0 = tj * 128;
    const int nk = C / 16;
    const int a_sub = 2 * ti + (wm >> 1);
    const int wgt = a_sub > tj ? 0 : (a_sub < tj ? 2 : 1);  // 0: below the diagonal (not written), 2: also mirrored

    const char* baseH = reinterpret_cast<const char*>(vph) + (int64_t)b * hw * C * 2;
    const char* baseL = reinterpret_cast<const char*>(vpl) + (int64_t)b * hw * C * 2;
    // wave w copies KiB (w & 3) of blocks (w >> 2), 2 + (w >> 2), 4 + (w >> 2): waves 0-3 -> Ah0, Al0, Bh; 4-7 -> Ah1, Al1, Bl
    const int wq = wave >> 2;
    const int64_t oA = ((int64_t)(2 * ti + wq) * nk) * GY_BLK + (wave & 3) * 1024;
    const int64_t oB = ((int64_t)tj * nk) * GY_BLK + (wave & 3) * 1024;
    const char* src0 = baseH + oA;
    const char* src1 = baseL + oA;
    const char* src2 = (wq ? baseL : baseH) + oB;
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)gy_smem);
    const uint32_t voff = (uint32_t)lane * 16;
    const uint32_t pw = (uint32_t)(wq * GY_BLK + (wave & 3) * 1024);
    auto stage = [&](int kc, int slot) __attribute__((always_inline)) {
        const int64_t ko = (int64_t)kc * GY_BLK;
        const uint32_t m0b = lds0 + (uint32_t)(slot * GY_SLOT) + pw;
#define GY_PIECE(I, SRC)                                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"((SRC) + ko),             \
                 "s"(m0b + (uint32_t)((I)*2 * GY_BLK))                                                             \
                 : "memory")
        GY_PIECE(0, src0);
        GY_PIECE(1, src1);
        GY_PIECE(2, src2);
#undef GY_PIECE
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // fragment offsets inside a slot: unit hi of row R sits at hi ^ ((R >> 3) & 1)
    const int u = (hi ^ ((l31 >> 3) & 1)) * 16;
    const int rA = (wm * 64 + l31) * 32 + u, rB = 4 * GY_BLK + (wn * 64 + l31) * 32 + u;

    stage(0, 0);
    if (nk > 1) {
        stage(1, 1);
        gx_wait_barrier<3>();
    } else {
        gx_wait_barrier<0>();
    }
    int slot = 0;
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 2 < nk) stage(kc + 2, slot >= 1 ? slot - 1 : GY_NS - 1);  // the slot chunk kc - 1 was read from
        const char* Ls = gy_smem + slot * GY_SLOT;
        if (wgt) {  // (the waves below the diagonal of a diagonal tile only copy and synchronise: 3 % of the products)
            half8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8_t*>(Ls + rA + i * 1024);
                al[i] = *reinterpret_cast<const half8_t*>(Ls + 2 * GY_BLK + rA + i * 1024);
                bh[i] = *reinterpret_cast<const half8_t*>(Ls + rB + i * 1024);
                bl[i] = *reinterpret_cast<const half8_t*>(Ls + GY_BLK + rB + i * 1024);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jj], acc[i][jj], 0, 0, 0);
                }
        }
        if (kc + 1 < nk) {
            if (kc + 2 < nk)
                gx_wait_barrier<3>();
            else
                gx_wait_barrier<0>();
        }
        slot = slot == GY_NS - 1 ? 0 : slot + 1;
    }
    // ---- epilogue ---- (gram_sign_epilogue above: registers -> global, nothing shared)
    const float* tgt = target + ((int64_t)b * hw + p0 + wm * 64 + 4 * hi) * hw + q0 + wn * 64 + l31;
    const float lsum = gram_sign_epilogue<LOSS>(acc, tgt, hw, wgt, wm, wn, l31, hi, sgn_out, b, p0, q0, s_tiled);
    if (LOSS) {
        float* red = reinterpret_cast<float*>(gy_smem);
        const float tot = wave_sum((float)wgt * lsum);
        __syncthreads();  // (every wave is done reading the ring)
        if (lane == 0) red[wave] = tot;
        __syncthreads();
        if (tid == 0) atomicAdd(loss, red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// gram16z_kernel: the same product on 128 x 128 workgroup tiles, 4 waves (2 x 2, wave tiles 64 x 64), THREE workgroups
// per CU (slot = Ah Al Bh Bl = 16 KB, 3 slots = 48 KB), for launches whose 256 x 128 tiles do not fill the chip's 512
// workgroup slots twice: at (1280, 32^2) the 320 tiles of gram16y_kernel put two workgroups on 64 CUs and one on the
// other 192 -- the launch takes as long as the CUs with two; 576 half-size tiles on 768 slots spread evenly.  The price
// is 16 KB instead of 12 KB of operand copies per 48 MFMAs, which is why the big planes keep the 256-row form.
// Tiles (ti, tj >= ti) in 128-pixel units; a diagonal tile is computed and written whole, the others are also written
// transposed.  Every entry is the same sum in the same order as in gram16y_kernel (same chunks, same three products per
// step, the same choice of which of S[p][q] / S[q][p] is computed and which is the copy): the two forms are bit-identical.
// grid (tiles per plane, 1, B), 256 threads, dynamic LDS 3 * 16 KB.
// ------------------------------------------------------------------------------------------------
constexpr int GZ_SLOT = 4 * GY_BLK;  // Ah Al Bh Bl
constexpr int GZ_NS = 3;

static int gz_tiles_per_plane(int hw) {
    const int n = hw / 128;
    return n * (n + 1) / 2;
}

template <bool LOSS>
__global__ __launch_bounds__(256, 3) void gram16z_kernel(const half_t* __restrict__ vph, const half_t* __restrict__ vpl,
                                                         const float* __restrict__ target, int8_t* __restrict__ sgn_out,
                                                         float* __restrict__ loss, int C, int hw, int s_tiled) {
    extern __shared__ __attribute__((aligned(16))) char gy_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int total = gridDim.x * gridDim.z;
    if (total % 8 == 0) lin = (lin % 8) * (total / 8) + lin / 8;  // XCD-contiguous ranges (as gram16y_kernel)
    const int b = lin / gridDim.x;
    int ti = 0, tj;
    {
        int idx = lin % gridDim.x;
        const int n = hw / 128;
        while (idx >= n - ti) {
            idx -= n - ti;
            ++ti;
        }
        tj = ti + idx;
    }
    const int p0 = ti * 128, q0 = tj * 128;
    const int nk = C / 16;
    const bool mirrored = ti < tj;

    const char* baseH = reinterpret_cast<const char*>(vph) + (int64_t)b * hw * C * 2;
    const char* baseL = reinterpret_cast<const char*>(vpl) + (int64_t)b * hw * C * 2;
    // wave w copies KiB w of each of the four blocks
    const int64_t oA = ((int64_t)ti * nk) * GY_BLK + wave * 1024;
    const int64_t oB = ((int64_t)tj * nk) * GY_BLK + wave * 1024;
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)gy_smem);
    const uint32_t voff = (uint32_t)lane * 16;
    auto stage = [&](int kc, int slot) __attribute__((always_inline)) {
        const int64_t ko = (int64_t)kc * GY_BLK;
        const uint32_t m0b = lds0 + (uint32_t)(slot * GZ_SLOT + wave * 1024);
#define GZ_PIECE(I, SRC)                                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"((SRC) + ko),             \
                 "s"(m0b + (uint32_t)((I)*GY_BLK))                                                                 \
                 : "memory")
        GZ_PIECE(0, baseH + oA);
        GZ_PIECE(1, baseL + oA);
        GZ_PIECE(2, baseH + oB);
        GZ_PIECE(3, baseL + oB);
#undef GZ_PIECE
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // fragment offsets inside a slot: unit hi of row R sits at hi ^ ((R >> 3) & 1)
    const int u = (hi ^ ((l31 >> 3) & 1)) * 16;
    const int rA = (wm * 64 + l31) * 32 + u, rB = 2 * GY_BLK + (wn * 64 + l31) * 32 + u;

    stage(0, 0);
    if (nk > 1) {
        stage(1, 1);
        gx_wait_barrier<4>();
    } else {
        gx_wait_barrier<0>();
    }
    int slot = 0;
    for (int kc = 0; kc < nk; ++kc) {
        const char* Ls = gy_smem + slot * GZ_SLOT;
        half8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const half8_t*>(Ls + rA + i * 1024);
            al[i] = *reinterpret_cast<const half8_t*>(Ls + GY_BLK + rA + i * 1024);
            bh[i] = *reinterpret_cast<const half8_t*>(Ls + rB + i * 1024);
            bl[i] = *reinterpret_cast<const half8_t*>(Ls + GY_BLK + rB + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jj], acc[i][jj], 0, 0, 0);
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jj], acc[i][jj], 0, 0, 0);
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jj], acc[i][jj], 0, 0, 0);
            }
        if (kc + 2 < nk) stage(kc + 2, slot >= 1 ? slot - 1 : GZ_NS - 1);  // the slot chunk kc - 1 was read from
        if (kc + 1 < nk) {
            if (kc + 2 < nk)
                gx_wait_barrier<4>();
            else
                gx_wait_barrier<0>();
        }
        slot = slot == GZ_NS - 1 ? 0 : slot + 1;
    }
    // ---- epilogue ---- (gram_sign_epilogue above: registers -> global, nothing shared)
    const float* tgt = target + ((int64_t)b * hw + p0 + wm * 64 + 4 * hi) * hw + q0 + wn * 64 + l31;
    const int wgt = mirrored ? 2 : 1;
    const float lsum = gram_sign_epilogue<LOSS>(acc, tgt, hw, wgt, wm, wn, l31, hi, sgn_out, b, p0, q0, s_tiled);
    if (LOSS) {
        float* red = reinterpret_cast<float*>(gy_smem);
        const float tot = wave_sum((float)wgt * lsum);
        __syncthreads();  // (every wave is done reading the ring)
        if (lane == 0) red[wave] = tot;
        __syncthreads();
        if (tid == 0) atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// dV^T = alpha * V^T S on fp16 MFMA with V split into two halfs:  V = Vh + Vl,  |V| <= 1, so the pair
// carries V to an absolute 2^-25 -- fp32 class -- and S in {-1,0,1} is exact in fp16; every product
// is exact in the fp32 accumulator.  2 x v_mfma_f32_32x32x16_f16 replace 8 x v_mfma_f32_32x32x2_f32:
// 1/8 of the matrix-pipe time of sv_kernel.  Both operands are read k-contiguous: A = rows c of
// V^T (k = pixel q), B = rows p of the SYMMETRIC sign matrix (S[q][p] = S[p][q]).
// Block tile 128 (c) x 128 (p), K chunk 32, LDS rows of 64 B + 16 B pad (conflict-free ds_read_b128).
// Requires hw % 16 == 0 (16-byte aligned rows); other sizes use sv_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int SV_GT = 128;         // block tile of sv16_kernel
constexpr int SK = 32;             // K chunk (pixels)
constexpr int SROW = SK * 2 + 16;  // LDS bytes per tile row

__global__ __launch_bounds__(256) void sv16_kernel(const half_t* __restrict__ vh, const half_t* __restrict__ vl,
                                                    const int8_t* __restrict__ sgn_in, float* __restrict__ dvt,
                                                    float* __restrict__ dotp, int C, int hw, float alpha) {
    __shared__ __attribute__((aligned(16))) char lds[2][3][SV_GT * SROW];  // [stage][Vh, Vl, S][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * SV_GT, p0 = blockIdx.x * SV_GT;
    const half_t* vhb = vh + (int64_t)b * C * hw;
    const half_t* vlb = vl + (int64_t)b * C * hw;
    const int8_t* sb = sgn_in + (int64_t)b * hw * hw;

    // staging: V tiles 128 rows x 4 chunks of 8 halfs -> 2 chunks per thread and array; S tile 128 rows x
    // 2 chunks of 16 int8 -> 1 chunk per thread, widened to 16 halfs when written to LDS
    uint4 rvh[2], rvl[2], rs;
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = tid + i * 256;
            const int row = ch >> 2, kc = ch & 3;
            const int c = c0 + row, k = k0 + kc * 8;
            uint4 a = make_uint4(0, 0, 0, 0), l = a;
            if (c < C && k < hw) {
                a = *reinterpret_cast<const uint4*>(vhb + (int64_t)c * hw + k);
                l = *reinterpret_cast<const uint4*>(vlb + (int64_t)c * hw + k);
            }
            rvh[i] = a;
            rvl[i] = l;
        }
        const int row = tid >> 1, kc = tid & 1;
        const int p = p0 + row, k = k0 + kc * 16;
        rs = make_uint4(0, 0, 0, 0);
        if (p < hw && k < hw) rs = *reinterpret_cast<const uint4*>(sb + (int64_t)p * hw + k);
    };
    auto store = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = tid + i * 256;
            const int row = ch >> 2, kc = ch & 3;
            *reinterpret_cast<uint4*>(&lds[st][0][row * SROW + kc * 16]) = rvh[i];
            *reinterpret_cast<uint4*>(&lds[st][1][row * SROW + kc * 16]) = rvl[i];
        }
        const int row = tid >> 1, kc = tid & 1;
        // 16 sign bytes -> 16 halfs: each byte IS the high byte of its fp16 value (sign_byte): two v_perm_b32 per dword
        const unsigned w4[4] = {rs.x, rs.y, rs.z, rs.w};
        u32x4 h0, h1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            h0[2 * e] = __builtin_amdgcn_perm(0u, w4[e], 0x010c000cu);          // [b1 0 b0 0]
            h0[2 * e + 1] = __builtin_amdgcn_perm(0u, w4[e], 0x030c020cu);      // [b3 0 b2 0]
            h1[2 * e] = __builtin_amdgcn_perm(0u, w4[2 + e], 0x010c000cu);
            h1[2 * e + 1] = __builtin_amdgcn_perm(0u, w4[2 + e], 0x030c020cu);
        }
        *reinterpret_cast<u32x4*>(&lds[st][2][row * SROW + kc * 32]) = h0;
        *reinterpret_cast<u32x4*>(&lds[st][2][row * SROW + kc * 32 + 16]) = h1;
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (hw + SK - 1) / SK;
    load(0);
    store(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int st = kc & 1;
        if (kc + 1 < nk) load((kc + 1) * SK);
        const char* ah = &lds[st][0][0];
        const char* al = &lds[st][1][0];
        const char* bs = &lds[st][2][0];
#pragma unroll
        for (int ks = 0; ks < SK / 16; ++ks) {
            half8_t fa[2][2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + l31) * SROW + ks * 32 + hi * 16;
                fa[i][0] = *reinterpret_cast<const half8_t*>(ah + off);
                fa[i][1] = *reinterpret_cast<const half8_t*>(al + off);
                fb[i] = *reinterpret_cast<const half8_t*>(bs + (wn * 64 + i * 32 + l31) * SROW + ks * 32 + hi * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kc + 1 < nk) store(st ^ 1);
        __syncthreads();
    }
    // epilogue: dV^T, and (dotp) this workgroup's share of <V, dV> per pixel: the sum over its 128 channels.
    // The V values of the lane's 64 elements are loaded BEFORE the first store (round 6): with the two loads behind every
    // store hipcc waited with vmcnt(0) in front of each fmaf -- on this target that counter covers the stores as well, so the 64
    // (store, load, load) groups of a lane ran one behind the other's round trip: ~11 of the launch's 19 us on the 8 x 8
    // planes.  Same products, same accumulation order: identical bits.
    float vv[2][2][16];
    if (dotp) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = p0 + wn * 64 + ni * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = c0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    vv[mi][ni][r] = (row < C && col < hw)
                                        ? (float)vhb[(int64_t)row * hw + col] + (float)vlb[(int64_t)row * hw + col]
                                        : 0.f;
                }
            }
    }
    float dsum[2] = {0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = p0 + wn * 64 + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < C && col < hw) {
                    const int64_t o = ((int64_t)b * C + row) * hw + col;
                    const float val = acc[mi][ni][r] * alpha;
                    dvt[o] = val;
                    if (dotp) dsum[ni] = fmaf(val, vv[mi][ni][r], dsum[ni]);
                }
            }
        }
    if (dotp) {
        float* red = reinterpret_cast<float*>(&lds[0][0][0]);  // [2][128]; the last chunk's reads are behind the loop's barrier
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float s = dsum[ni];
            s += __shfl_xor(s, 32, 64);
            if (hi == 0) red[wm * 128 + wn * 64 + ni * 32 + l31] = s;
        }
        __syncthreads();
        if (tid < 128 && p0 + tid < hw) dotp[((int64_t)b * gridDim.y + blockIdx.y) * hw + p0 + tid] = red[tid] + red[128 + tid];
    }
}

// ------------------------------------------------------------------------------------------------
// The same product for the big planes (hw % 256 == 0, C % 128 == 0): 128 (c) x 256 (p) workgroup tiles, 8 waves of
// 64 x 64, two workgroups per CU.  Ablation of sv16_kernel (profiles/r02_attn_experiments.txt section 5): 45 % of its
// time is the staging work itself -- each thread pays 5 global loads + 6 ds_write_b128 (with the sign expansion) per 16
// MFMAs, and two chunks of register look-ahead do not help.  Here
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no VALU) into a ring of slots behind
//     counted vmcnt waits and one barrier per chunk (the protocol of proj.hip / attn.hip); with the registers that
//     frees (108) two workgroups share a CU, so a 2-slot ring (one chunk ahead) is enough: while one workgroup waits
//     for its chunk the other multiplies;
//   * S stays ONE BYTE per sign in LDS (the fp16 high byte, sign_byte): half the LDS bytes of the widened form, expanded
//     to packed halfs after the ds_read_b64 with two v_perm_b32 per dword.
// Round 4: LDS rows WITHOUT pad bytes (V 64 B, S 32 B): the producers store both operands swizzled -- 16-byte unit u
// of V row c at u ^ ((c >> 2) & 3), 8-byte unit u of S row p at u ^ ((p >> 3) & 3) -- so a slot is 24 linear 1 KiB DMA
// pieces (3 per wave; 32 with the pad chunks before: -25 % bytes and DMA issues per MFMA), fragment reads stay
// conflict-free (the padded S rows had 2-way conflicts: 25 % of the LDS cycles, profiles/r04_pmc_removed_gram16x_kernel.csv), and
// three slots (72 KB) fit twice per CU: chunks arrive two steps ahead.
// ------------------------------------------------------------------------------------------------
#ifndef FRESCO_SV_SPLIT_READS
#define FRESCO_SV_SPLIT_READS 1
#endif
constexpr int SB_TC = 128, SB_K = 32;
constexpr int SB_VROW = SB_K * 2, SB_SROW = SB_K;
constexpr int SB_NSLOT = 3;
template <int N_>
__device__ __forceinline__ void sb_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N_) : "memory");
}

// CT = channel rows per workgroup: 128 (8 waves as 2 x 4, wave tiles 64 x 64) for the full rounds of a launch, 64 (8 waves
// as 1 x 8, wave tiles 64 x 32) for the tiles of the last, partly filled round: 1280 tiles on 512 workgroup slots are
// 2.5 rounds -- run as 2 rounds of whole tiles + 1 round of half tiles (0.6 of the time) instead of 3 whole rounds.
// Tile t of the launch's list = (plane, pixel tile, channel tile), channel tile fastest (neighbours share their S rows).
// <V, dV> partials: two slots per channel tile (a whole tile writes its sum and a zero, half tiles one each).
template <int CT>
__global__ __launch_bounds__(512, 4) void sv16b_kernel(const half_t* __restrict__ vh, const half_t* __restrict__ vl,
                                                       const int8_t* __restrict__ sgn_in, float* __restrict__ dvt,
                                                       float* __restrict__ dotp, int C, int hw, float alpha, int tile_base) {
    constexpr int TP = 256, NS = SB_NSLOT;
    constexpr int WM = CT / 64, WN = 8 / WM, NJ = TP / (32 * WN);
    constexpr int VARR = CT * SB_VROW, SARR = TP * SB_SROW, SLOT = 2 * VARR + SARR;
    extern __shared__ __attribute__((aligned(16))) char sb_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nct = C / SB_TC, npt = hw / TP;
    int t, hf = 0;
    if (CT == 128) {
        // XCD-aware order: consecutive workgroup ids land on different XCDs (id % 8), each with its own L2: give every XCD a
        // CONTIGUOUS range of the tile list, so that workgroups sharing operand rows run on the same L2
        int g = blockIdx.x;
        const int G = gridDim.x;
        if (G % 8 == 0) g = (g % 8) * (G / 8) + g / 8;
        t = tile_base + g;
    } else {
        t = tile_base + (blockIdx.x >> 1);
        hf = blockIdx.x & 1;
    }
    const int b = t / (nct * npt);
    const int c0 = (t % nct) * SB_TC, p0 = ((t / nct) % npt) * TP;
    // operands are pre-tiled (sv_tiled_layout): per (channel tile, pixel chunk) 128 x 32 halfs, per (pixel tile, chunk) 256 x 32 bytes
    const char* vhb = reinterpret_cast<const char*>(vh + ((int64_t)b * nct + c0 / 128) * hw * 128);
    const char* vlb = reinterpret_cast<const char*>(vl + ((int64_t)b * nct + c0 / 128) * hw * 128);
    const char* sbp = reinterpret_cast<const char*>(sgn_in + ((int64_t)b * npt + p0 / 256) * hw * 256);
    const uint32_t lds0 =
        __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)sb_smem);

    // DMA, 1 KiB pieces: CT = 128: wave w copies KiB w of Vh, of Vl and of S (3 pieces); CT = 64: waves 0-3 KiB w of the Vh
    // half, waves 4-7 KiB w - 4 of the Vl half, every wave KiB w of S (2 pieces)
    constexpr int NPW = CT == 128 ? 3 : 2;
    const uint32_t voff = (uint32_t)lane * 16;
    const char* s_a = CT == 128 ? vhb + wave * 1024 : (wave < 4 ? vhb : vlb) + hf * VARR + (wave & 3) * 1024;
    const char* s_l = vlb + wave * 1024;  // (CT = 128 only)
    const char* s_s = sbp + wave * 1024;
    const uint32_t o_a = CT == 128 ? (uint32_t)(wave * 1024) : (uint32_t)((wave < 4 ? 0 : VARR) + (wave & 3) * 1024);
    auto stage = [&](int kc, int slot) __attribute__((always_inline)) {
        const uint32_t m0b = lds0 + (uint32_t)(slot * SLOT);
#define SB_PIECE(OFF, SRC)                                                                                        \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(SRC), "s"(m0b + (uint32_t)(OFF)) \
                 : "memory")
        SB_PIECE(o_a, s_a + (int64_t)kc * (SB_TC * SB_VROW));
        if (CT == 128) SB_PIECE(VARR + wave * 1024, s_l + (int64_t)kc * (SB_TC * SB_VROW));
        SB_PIECE(2 * VARR + wave * 1024, s_s + (int64_t)kc * SARR);
#undef SB_PIECE
    };
    auto wait_barrier = [&](int keep) __attribute__((always_inline)) {  // keep = newer slots that may stay in flight
        if (keep == 0)
            sb_wait_barrier<0>();
        else
            sb_wait_barrier<NPW>();
    };

    floatx16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = hw / SB_K;
    static_assert(NS == 3, "ring protocol below: chunks two steps ahead");
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    wait_barrier(nk > 1 ? 1 : 0);
    // fragment addresses: V row R (64 B): unit (ks*2 + hi) at ^((R >> 2) & 3); S row P (32 B): 8-byte unit (ks*2 + hi) at ^((P >> 3) & 3)
    const int swv = (l31 >> 2) & 3, sws = (l31 >> 3) & 3;
    const int ra = (wm * 64 + l31) * SB_VROW, rs = 2 * VARR + (wn * (32 * NJ) + l31) * SB_SROW;
    int slot = 0;
    for (int kc = 0; kc < nk; ++kc) {
        const char* base = sb_smem + slot * SLOT;
#pragma unroll
        for (int ks = 0; ks < SB_K / 16; ++ks) {
            half8_t fa[2][2], fb[NJ];
            const int ua = ((ks * 2 + hi) ^ swv) * 16, us = ((ks * 2 + hi) ^ sws) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][0] = *reinterpret_cast<const half8_t*>(base + ra + i * 32 * SB_VROW + ua);
                fa[i][1] = *reinterpret_cast<const half8_t*>(base + VARR + ra + i * 32 * SB_VROW + ua);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // (round 5: hipcc fused the NJ reads of a k-step into ONE ds_read2st64_b64 -- which the LDS services in
                // 16-lane groups against 32 banks: 8 cycles + 2-way conflicts on these 32-byte rows, SQ_LDS_BANK_CONFLICT = 24 %
                // of the LDS cycles in profiles/r05_pmc_opt_C640_h64.csv -- where two ds_read_b64 take 2 conflict-free cycles
                // each (32-lane groups, 64 banks: the layout the swizzle was designed for).  The laundered offset keeps them apart.)
                int sj = j * 32 * SB_SROW;
                if (FRESCO_SV_SPLIT_READS && j > 0) asm volatile("" : "+v"(sj));
                const u32x2 raw = *reinterpret_cast<const u32x2*>(base + rs + sj + us);
                u32x4 w;
                w[0] = __builtin_amdgcn_perm(0u, raw[0], 0x010c000cu);
                w[1] = __builtin_amdgcn_perm(0u, raw[0], 0x030c020cu);
                w[2] = __builtin_amdgcn_perm(0u, raw[1], 0x010c000cu);
                w[3] = __builtin_amdgcn_perm(0u, raw[1], 0x030c020cu);
                fb[j] = __builtin_bit_cast(half8_t, w);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j], acc[i][j], 0, 0, 0);
                }
        }
        // (the copies of chunk kc + 2 are requested BEHIND the products of chunk kc, where an LDS-DMA instruction is cheapest to
        // issue: 482 -> 471 us at (640, 64^2); the same move in gram16y_kernel costs 7 us, in gram16z_kernel it gains 1.5)
        if (kc + 2 < nk) stage(kc + 2, slot >= 1 ? slot - 1 : NS - 1);  // the slot chunk kc - 1 was read from
        if (kc + 1 < nk) wait_barrier(kc + 2 < nk ? 1 : 0);
        slot = slot == NS - 1 ? 0 : slot + 1;
    }
    // epilogue: dV^T, and (dotp) this workgroup's share of <V, dV> per pixel: the sum over its channels, V = Vh + Vl
    // re-read from the tiled copies (L2-resident: this workgroup has just streamed them) THROUGH LDS: the ring is free,
    // every wave copies the rows of its own tile -- 32 channel rows x NJ pixel chunks x (hi, lo) = NJ x 4 KB, as linear
    // 1 KiB LDS-DMA pieces -- into a region of its own (no barrier: the wave that copies is the wave that reads), picks
    // its values with ds_read_u16 through the same swizzle, and only then issues the dV stores (stores count in vmcnt:
    // they must not sit in front of the copies).  Before: 128 two-byte global gathers per lane -- 496 -> 469 us at
    // (640, 64^2), 99 -> 85 at (1280, 32^2), 24 -> 17 at (1280, 16^2); same products in the same order, same bits
    // (profiles/r04_ab_opt_forms.txt).
    float dsum[NJ];
#pragma unroll
    for (int ni = 0; ni < NJ; ++ni) dsum[ni] = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= alpha;  // (in place: both phases below read the scaled value)
    if (dotp) {
        __syncthreads();  // the ring is free
        const uint32_t mybase = lds0 + (uint32_t)(wave * (NJ * 4096));
        const char* myrd = sb_smem + wave * (NJ * 4096);
        const int ch0 = (p0 >> 5) + wn * NJ;  // first pixel chunk of the wave's columns
        // a lane's value of accumulator row r sits at row lr = (r & 3) + 8 (r >> 2) + 4 hi of the 32, unit ((l31 >> 3) ^ (lr >> 2)) & 3
        // of the 64-byte row: two lane offsets (r >> 2 even / odd), the rest are immediates
        const int lb0 = hi * 256 + ((((l31 >> 3) ^ hi) & 3) << 4) + (l31 & 7) * 2, lb1 = lb0 ^ 32;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int rl0 = hf * 64 + wm * 64 + mi * 32;  // first of the 32 channel rows (inside the 128-channel tile)
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
                for (int ar = 0; ar < 2; ++ar)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const char* src = (ar ? vlb : vhb) + ((int64_t)(ch0 + ni) * 128 + rl0) * SB_VROW + pc * 1024;
                        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src),
                                     "s"(mybase + (uint32_t)(ni * 4096 + ar * 2048 + pc * 1024))
                                     : "memory");
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int ni = 0; ni < NJ; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = (((r >> 2) & 1) ? lb1 : lb0) + ni * 4096 + ((r & 3) + 8 * (r >> 2)) * SB_VROW;
                    const float vv = (float)*reinterpret_cast<const half_t*>(myrd + off) +
                                     (float)*reinterpret_cast<const half_t*>(myrd + off + 2048);
                    dsum[ni] = fmaf(acc[mi][ni][r], vv, dsum[ni]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next copies overwrite the rows just read)
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni) {
            const int col = p0 + wn * (32 * NJ) + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = hf * 64 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;  // row inside the 128-channel tile
                dvt[((int64_t)b * C + c0 + rl) * hw + col] = acc[mi][ni][r];
            }
        }
    if (dotp) {
        __syncthreads();  // the ring is free
        float* red = reinterpret_cast<float*>(sb_smem);  // [WM][TP]
#pragma unroll
        for (int ni = 0; ni < NJ; ++ni) {
            float s = dsum[ni];
            s += __shfl_xor(s, 32, 64);
            if (hi == 0) red[wm * TP + wn * (32 * NJ) + ni * 32 + l31] = s;
        }
        __syncthreads();
        if (tid < TP) {
            float* dp = dotp + ((int64_t)b * (2 * nct) + 2 * (c0 / SB_TC)) * hw + p0 + tid;
            if (CT == 128) {  // (one slot per 64 channels in both tile shapes: the Adam kernel adds them in the same order)
                dp[0] = red[tid];
                dp[hw] = red[TP + tid];
            } else {
                dp[(int64_t)hf * hw] = red[tid];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void launch_sv16_plain(const half_t* vh, const half_t* vl, const int8_t* ssign, float* dvt, float* dotp, int B, int C,
                       int hw, float alpha, hipStream_t st) {
    const int nt = (hw + SV_GT - 1) / SV_GT;
    hipLaunchKernelGGL(sv16_kernel, dim3(nt, (C + SV_GT - 1) / SV_GT, B), dim3(256), 0, st, vh, vl, ssign, dvt, dotp, C,
                       hw, alpha);
}

void opt_fast_begin(const OptWs& w, const float* cs, int planes, int C, int hw, int Bg, hipStream_t st) {
    int K, NPART, NPB;
    fast_slices(C, Bg, hw, ADAM_K, &K, &NPART, &NPB);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(hw / 64, NPB, planes), dim3(256), 0, st, cs, w.part, C, hw, K, NPART);
}

void opt_fast_closure(const OptWs& w, float* cs, const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                      const float* bwd_occ, const float* target, int nck, int C, int h, int wd, float intra_weight,
                      int has_t, int mode, float* gout, float* loss, AdamArgs a, hipStream_t st, const TLayout& Lin,
                      int Bg, const FastSync* sync) {
    const int hw = h * wd;
    TLayout L = Lin;
    if (!has_t) L = TLayout{Lin.n_loc, Lin.n_loc, 1, nullptr, nullptr};
    const int planes = nck * L.n_loc;
    int K, NPART, NPB, Kp, NPARTp, NPBp;
    fast_slices(C, Bg, hw, ADAM_K, &K, &NPART, &NPB);       // adam (and the partial sums of squares it leaves behind)
    fast_slices(C, Bg, hw, PREP_K, &Kp, &NPARTp, &NPBp);    // prep
    const int NCT = (C + 127) / 128;  // channel tiles of the S V kernels
    const bool cm_tiled = sv_tiled_layout(hw, C);
    const bool small = hw <= 256 && C % 32 == 0;
    // (round 6 experiment: FRESCO_GRAM_TILED_MIN_HW=256 sends the 16 x 16 planes to the DMA-staged 128 x 128 tile kernel instead
    // of gram16s_kernel, whose row-strided 16-byte fragment loads sustain ~9 B/clk/CU; read per call: the tests switch it)
    const char* mh_env = getenv("FRESCO_GRAM_TILED_MIN_HW");
    const int min_hw = mh_env ? atoi(mh_env) : 512;
    const bool big = gram_x_layout(hw, C, min_hw > 0 ? min_hw : 512);
    const float kscale = 2.f / ((float)Bg * (float)C * (float)hw);
    const int parts = sync ? sync->parts : 3;
    const int halo_split = (sync && sync->halo_split && has_t && !L.circular) ? 1 : 0;
    auto launch_prep = [&](int phase) {
        ProfScope ps(FRESCO_PROF_OPT_TSIGN, planes, C, hw, phase, st);
        PrepArgs pa;
        pa.cs = cs;
        pa.part = w.part;
        pa.nrm = w.nrm;
        pa.vh = w.vh;
        pa.vl = w.vl;
        pa.vph = w.vph;
        pa.vpl = w.vpl;
        pa.bwd_flow = bwd_flow;
        pa.fwd_flow = fwd_flow;
        pa.bwd_occ = bwd_occ;
        pa.fwd_occ = fwd_occ;
        pa.sgn1 = w.sgn1;
        pa.sgn2 = w.sgn2;
        pa.loss = loss;
        pa.L = L;
        pa.C = C;
        pa.h = h;
        pa.w = wd;
        pa.K = Kp;
        pa.NPART = NPARTp;
        pa.NPB = NPB;  // (the partial sums of squares it adds: adam's blocks)
        pa.has_t = has_t;
        pa.pm_tiled = big ? 1 : 0;
        pa.cm_tiled = cm_tiled ? 1 : 0;
        pa.phase = phase;
        const int nz = nck * (has_t ? L.n_pairs : L.n_loc);
        hipLaunchKernelGGL(opt_prep_kernel, dim3(hw / 64, NPBp, nz), dim3(256), 0, st, pa);
    };
    if (parts & 1) {
    launch_prep(halo_split ? 1 : 0);
    float* gloss = loss ? loss + 1 : nullptr;
    if (sync && sync->wait_before_gram) (void)hipStreamWaitEvent(st, sync->wait_before_gram, 0);
    {
        ProfScope ps(FRESCO_PROF_OPT_GRAM, planes, C, hw, 0, st);
        if (big) {
            // launches whose 256 x 128 tiles would not fill the 512 workgroup slots twice take the 128 x 128 form (three
            // workgroups per CU; same bits): (1280, 32^2) 95 -> 83 us.  FRESCO_GRAM_Z=0 keeps the 256-row form (tests)
            const char* z_env = getenv("FRESCO_GRAM_Z");  // (read per call)
            const bool zform = !(z_env && z_env[0] == '0') && gx_tiles_per_plane(hw) * Bg < 1024;
            if (zform) {
                constexpr int ldsz = GZ_NS * GZ_SLOT;
                static const bool oncez = [] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16z_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsz);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16z_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, ldsz);
                    return true;
                }();
                (void)oncez;
                const dim3 gridz(gz_tiles_per_plane(hw), 1, planes);
                if (gloss)
                    hipLaunchKernelGGL(gram16z_kernel<true>, gridz, dim3(256), ldsz, st, w.vph, w.vpl, target, w.ssign, gloss, C,
                                       hw, cm_tiled ? 1 : 0);
                else
                    hipLaunchKernelGGL(gram16z_kernel<false>, gridz, dim3(256), ldsz, st, w.vph, w.vpl, target, w.ssign, gloss, C,
                                       hw, cm_tiled ? 1 : 0);
            } else {
                constexpr int lds = GY_NS * GY_SLOT;
                static const bool once = [] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16y_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16y_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    return true;
                }();
                (void)once;
                const dim3 grid(gx_tiles_per_plane(hw), 1, planes);
                if (gloss)
                    hipLaunchKernelGGL(gram16y_kernel<true>, grid, dim3(512), lds, st, w.vph, w.vpl, target, w.ssign, gloss, C,
                                       hw, cm_tiled ? 1 : 0);
                else
                    hipLaunchKernelGGL(gram16y_kernel<false>, grid, dim3(512), lds, st, w.vph, w.vpl, target, w.ssign, gloss, C,
                                       hw, cm_tiled ? 1 : 0);
            }
        } else if (small) {
            const int nt = hw / 64;
            // (the wave count is the split of the contraction, i.e. part of the arithmetic: chosen by the size of the WHOLE
            // problem, so that one CFG half alone, or a rank's frame shard, rounds exactly as the undivided batch does)
            // 8 waves (an 8-way split of K) up to 16 x 16 planes at batch 16: the launch is a latency chain of K steps
            // FRESCO_GRAM_COOP (read per call; tests): "0" = gram16s_kernel's lane-per-row fragment loads (rounds 3-5),
            // default = gram16c_kernel's coalesced cooperative staging (same bits)
            const char* co_env = getenv("FRESCO_GRAM_COOP");
            const bool coop = !(co_env && co_env[0] == '0');
            const char* sp_env = getenv("FRESCO_GRAM_SPLIT_WG");  // (read per call: "0" keeps the one-launch form; tests)
            if (nt * nt * Bg < 512 && nt * nt * planes <= 64 && w.gpart && !gloss && hw <= 64 && !(sp_env && sp_env[0] == '0')) {
                // one 64 x 64 tile per plane: every split-K slice on a CU of its own, then the ordered sum (same bits)
                hipLaunchKernelGGL(gram16sp_kernel<8>, dim3(nt * nt, 8, planes), dim3(64), 0, st, w.vph, w.vpl, w.gpart, C, hw);
                hipLaunchKernelGGL(gram16sr_kernel, dim3(nt * nt, 1, planes), dim3(256), 0, st, w.gpart, target, w.ssign, 8, hw,
                                   cm_tiled ? 1 : 0);
            } else if (coop && nt * nt * Bg < 512) {
                constexpr int lds = GramCCfg<8>::LDS;
                static const bool once = [] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16c_kernel<8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    return true;
                }();
                (void)once;
                hipLaunchKernelGGL(gram16c_kernel<8>, dim3(nt * nt, 1, planes), dim3(512), lds, st, w.vph, w.vpl, target,
                                   w.ssign, gloss, C, hw, cm_tiled ? 1 : 0);
            } else if (coop) {
                constexpr int lds = GramCCfg<4>::LDS;
                static const bool once = [] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16c_kernel<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    return true;
                }();
                (void)once;
                hipLaunchKernelGGL(gram16c_kernel<4>, dim3(nt * nt, 1, planes), dim3(256), lds, st, w.vph, w.vpl, target,
                                   w.ssign, gloss, C, hw, cm_tiled ? 1 : 0);
            } else if (nt * nt * Bg < 512) {
                constexpr int lds = 8 * 64 * GS_RS * 4;
                static const bool once = [] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16s_kernel<8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    return true;
                }();
                (void)once;
                hipLaunchKernelGGL(gram16s_kernel<8>, dim3(nt * nt, 1, planes), dim3(512), lds, st, w.vph, w.vpl, target,
                                   w.ssign, gloss, C, hw, cm_tiled ? 1 : 0);
            } else {
                constexpr int lds = 4 * 64 * GS_RS * 4;
                static const bool once = [] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16s_kernel<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    return true;
                }();
                (void)once;
                hipLaunchKernelGGL(gram16s_kernel<4>, dim3(nt * nt, 1, planes), dim3(256), lds, st, w.vph, w.vpl, target,
                                   w.ssign, gloss, C, hw, cm_tiled ? 1 : 0);
            }
        } else {
            launch_gram16_plain(w.vph, w.vpl, target, w.ssign, gloss, planes, C, hw, st);
        }
    }
    if (sync && sync->record_after_gram) (void)hipEventRecord(sync->record_after_gram, st);
    const float coef = intra_weight / ((float)Bg * (float)hw * (float)hw);
    {
        ProfScope ps(FRESCO_PROF_OPT_SV, planes, C, hw, 0, st);
        if (cm_tiled) {
            constexpr int lds128 = SB_NSLOT * (2 * 128 * SB_VROW + 256 * SB_SROW), lds64 = SB_NSLOT * (2 * 64 * SB_VROW + 256 * SB_SROW);
            static const bool once = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sv16b_kernel<128>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sv16b_kernel<64>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds64);
                return true;
            }();
            (void)once;
            // whole tiles for the full rounds of the chip's 2 x 256 workgroup slots; the tiles of a last round that would
            // fill at most a quarter of the slots (or a launch smaller than one round) run as two half tiles each: measured
            // 116 -> 102 us at (1280, 32^2), 28 -> 21 us at (1280, 16^2); at (640, 64^2) -- 2.5 rounds -- the half-tile round
            // does not pay (514 -> 534 us).  FRESCO_OPT_SVTAIL=0: whole tiles only
            const char* tail_env = getenv("FRESCO_OPT_SVTAIL");  // (read per call: the tests switch it)
            const int tail_split = (tail_env && tail_env[0] == '0') ? 0 : 1;
            const int tiles = (hw / 256) * (C / SB_TC) * planes, slots = 512;
            int rem = tiles % slots;
            if (!tail_split || (rem > slots / 4 && tiles > slots)) rem = 0;
            if (tiles <= slots && tiles > slots / 2) rem = 0;  // (more than half a round of whole tiles: leave it)
            const int whole = tiles - rem;
            if (whole > 0)
                hipLaunchKernelGGL(sv16b_kernel<128>, dim3(whole), dim3(512), lds128, st, w.vh, w.vl, w.ssign, w.dvt, w.dotp, C,
                                   hw, 2.f * coef, 0);
            if (rem > 0)
                hipLaunchKernelGGL(sv16b_kernel<64>, dim3(2 * rem), dim3(512), lds64, st, w.vh, w.vl, w.ssign, w.dvt, w.dotp, C,
                                   hw, 2.f * coef, whole);
        } else {
            launch_sv16_plain(w.vh, w.vl, w.ssign, w.dvt, w.dotp, planes, C, hw, 2.f * coef, st);
        }
    }
    if (sync && sync->record_after_sv) (void)hipEventRecord(sync->record_after_sv, st);
    }  // parts & 1
    if (!(parts & 2)) return;
    if (halo_split) launch_prep(2);  // (the halo frames have arrived: signs of the two boundary pairs)
    if (sync && sync->wait_before_adam) (void)hipStreamWaitEvent(st, sync->wait_before_adam, 0);
    {
        ProfScope ps(FRESCO_PROF_OPT_ADAM, planes, C, hw, 0, st);
        AdamKArgs ka;
        ka.cs = cs;
        ka.m = w.m;
        ka.v2 = w.v;
        ka.tg = TGradArgs{w.sgn1, w.sgn2, bwd_occ, fwd_occ, w.rowptr, w.src, w.wgt, L, kscale};
        ka.dvt = w.dvt;
        ka.nrm = w.nrm;
        ka.dotp = w.dotp;
        ka.part = w.part;
        ka.gout = gout;
        ka.C = C;
        ka.hw = hw;
        ka.K = K;
        ka.NPART = NPART;
        ka.NCT = cm_tiled ? 2 * NCT : NCT;  // (sv16b writes two <V, dV> slots per channel tile)
        ka.has_t = has_t;
        ka.has_s = 1;
        ka.mode = mode;
        ka.a = a;
        if ((int64_t)Bg * C * hw >= (int64_t)16 << 20)  // (the WHOLE batch decides, not this launch's share of it)
            hipLaunchKernelGGL(opt_adam_kernel<true>, dim3(hw / 64, NPB, planes), dim3(256), 0, st, ka);
        else
            hipLaunchKernelGGL(opt_adam_kernel<false>, dim3(hw / 64, NPB, planes), dim3(256), 0, st, ka);
    }
}

}  // namespace fresco
