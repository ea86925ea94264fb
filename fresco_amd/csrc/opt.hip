// FRESCO feature optimisation (reference: src/diffusion_hacked.py:416-488), autograd-free, fp32.
//
//   L(cs) = 2*mean(|(c2 - W_b c1) mb| + |(c1 - W_f c2) mf|) + w*mean|V V^T - T|
//   c1 = cs[f], c2 = cs[f+1 mod N], mb = 1-occ_b, mf = 1-occ_f, V = rows of cs (hw x C) L2-normalised.
//
// Gradients, per CFG half, frame f, with k = 2/(B*C*hw):
//   s1[f] = sign((c2 - W_b[f] c1) mb[f]) mb[f] k,   s2[f] = sign((c1 - W_f[f] c2) mf[f]) mf[f] k
//   dL/dcs[f] = s2[f] + s1[f-1] - W_b[f]^T s1[f] - W_f[f-1]^T s2[f-1]
//   S = sign(V V^T - T) w/(B hw^2);  dV = (S + S^T) V = 2 S V (T symmetric);  dX = (dV - V <V,dV>)/|X|
//
// Kernels:
//   csr_build      one block per (direction, frame): transposes the 4-tap bilinear matrix into CSR so
//                  that W^T s is a deterministic gather (rows sorted by source pixel); entry weight
//                  already carries mask[src]*k.  Built once per optimize_feature call.
//   temporal_sign  signs of both residuals as int8 (HBM-bound, 4-tap gathers, taps shared by channels)
//   (TGradPixel)   dL/dcs of the temporal term from the int8 signs + CSR gathers, evaluated inside adam_update
//   colnorm        |X[p]| and V^T (C x hw: the NCHW plane layout IS V^T, so both MFMA operands of
//                  V V^T are read with lanes along consecutive pixels)
//   gram           128x128 tiles of V V^T on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain);
//                  epilogue writes sign(G - T) as int8 (or G itself for the Gram target)
//   sv             dV^T = 2 coef V^T S on the same MFMA (S in {-1,0,1}, exact)
//   (fp16-split forms, the ones the SD-1.5 shapes run: gram16w / gram16 and sv16b / sv16 further down -- three resp. two
//   v_mfma_f32_32x32x16_f16 per fp32-accurate product, operands by LDS-DMA from pre-tiled copies, 16 waves per CU)
//   adam_update    temporal gradient + norm backward + Adam step, fused, fp32 state
#include "opt_shared.h"
#include <new>
#include <stdlib.h>
#include <atomic>

namespace fresco {



// ------------------------------------------------------------------------------------------------
// CSR of W^T.  grid (N, 2): blockIdx.y = 0 -> bwd flow (samples c1), 1 -> fwd flow (samples c2).
// rowptr: [2][N][hw+1], cursor: [2][N][hw] scratch, src: [2][N][4hw], wgt: [2][N][4hw]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void csr_build_kernel(const float* __restrict__ bwd_flow,
                                                          const float* __restrict__ fwd_flow,
                                                          const float* __restrict__ bwd_occ,
                                                          const float* __restrict__ fwd_occ,
                                                          int* __restrict__ rowptr, int* __restrict__ cursor,
                                                          int* __restrict__ src, float* __restrict__ wgt,
                                                          int N, int h, int w, float kscale) {
    const int hw = h * w;
    const int f = blockIdx.x, dir = blockIdx.y;
    const float* flow = (dir == 0 ? bwd_flow : fwd_flow) + (int64_t)f * 2 * hw;
    const float* occ = (dir == 0 ? bwd_occ : fwd_occ) + (int64_t)f * hw;
    int* rp = rowptr + (int64_t)(dir * N + f) * (hw + 1);
    int* cur = cursor + (int64_t)(dir * N + f) * hw;
    int* sp = src + (int64_t)(dir * N + f) * 4 * hw;
    float* wp = wgt + (int64_t)(dir * N + f) * 4 * hw;
    const int tid = threadIdx.x;
    __shared__ int scan_buf[1024];
    __shared__ int carry;

    for (int i = tid; i < hw; i += 1024) cur[i] = 0;
    __syncthreads();
    // count entries per target
    for (int p = tid; p < hw; p += 1024) {
        const OTaps t = otaps(flow[p], flow[hw + p], p % w, p / w, h, w);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t.w[j] != 0.f) atomicAdd(&cur[t.idx[j]], 1);
    }
    __syncthreads();
    // exclusive scan -> rowptr
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < hw; base += 1024) {
        const int i = base + tid;
        const int val = i < hw ? atomicAdd(&cur[i], 0) : 0;  // counts were built by L2 atomics
        scan_buf[tid] = val;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int add = tid >= off ? scan_buf[tid - off] : 0;
            __syncthreads();
            scan_buf[tid] += add;
            __syncthreads();
        }
        const int incl = scan_buf[tid];
        const int c0 = carry;
        if (i < hw) rp[i] = c0 + incl - val;
        __syncthreads();
        if (tid == 1023) carry = c0 + incl;
        __syncthreads();
    }
    if (tid == 0) rp[hw] = carry;
    __syncthreads();
    for (int i = tid; i < hw; i += 1024) cur[i] = rp[i];
    __syncthreads();
    // fill (arbitrary order within a row)
    for (int p = tid; p < hw; p += 1024) {
        const OTaps t = otaps(flow[p], flow[hw + p], p % w, p / w, h, w);
        const float mk = (1.f - occ[p]) * kscale;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t.w[j] != 0.f) {
                const int pos = atomicAdd(&cur[t.idx[j]], 1);
                sp[pos] = p;
                wp[pos] = t.w[j] * mk;
            }
    }
    __syncthreads();
    // sort every row by source pixel -> deterministic summation order
    for (int r = tid; r < hw; r += 1024) {
        const int b = rp[r], e = rp[r + 1];
        for (int i = b + 1; i < e; ++i) {
            const int ks = sp[i];
            const float kw = wp[i];
            int j = i - 1;
            while (j >= b && sp[j] > ks) {
                sp[j + 1] = sp[j];
                wp[j + 1] = wp[j];
                --j;
            }
            sp[j + 1] = ks;
            wp[j + 1] = kw;
        }
    }
}


// grid (ceil(hw/256), ceil(C/OCPT), chunk*n_pairs).  sgn1/sgn2: (chunk*n_pairs, C, hw) int8; flows / occs
// are indexed by pair.  loss (optional): loss[0] += sum |r1| + |r2|  (unscaled; circular layout only)
__global__ __launch_bounds__(256) void temporal_sign_kernel(
    const float* __restrict__ cs, const float* __restrict__ bwd_flow, const float* __restrict__ fwd_flow,
    const float* __restrict__ bwd_occ, const float* __restrict__ fwd_occ, int8_t* __restrict__ sgn1,
    int8_t* __restrict__ sgn2, float* __restrict__ loss, TLayout L, int C, int h, int w) {
    const int hw = h * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z, ck = b / L.n_pairs, j = b % L.n_pairs;
    const int sa = j, sb = L.circular ? (j + 1) % L.n_loc : j + 1;
    const int c0 = blockIdx.y * OCPT, cend = min(c0 + OCPT, C);
    float lsum = 0.f;
    if (p < hw) {
        const float* fb = bwd_flow + (int64_t)j * 2 * hw;
        const float* ff = fwd_flow + (int64_t)j * 2 * hw;
        const OTaps tb = otaps(fb[p], fb[hw + p], p % w, p / w, h, w);
        const OTaps tf = otaps(ff[p], ff[hw + p], p % w, p / w, h, w);
        const float mb = 1.f - bwd_occ[(int64_t)j * hw + p];
        const float mf = 1.f - fwd_occ[(int64_t)j * hw + p];
        // the 8 signs of this thread's channel octet are ONE 8-byte word: signs live as [pair][C/8][hw][8] bytes, so a
        // thread writes (and the gradient code reads, per CSR entry) 8 bytes at once and lanes along p stay coalesced
        uint64_t w1 = 0, w2 = 0;
        for (int c = c0; c < cend; ++c) {
            const float* c1 = frame_plane(cs, L, ck, sa, c, C, hw);
            const float* c2 = frame_plane(cs, L, ck, sb, c, C, hw);
            const float r1 = (c2[p] - osample(c1, tb)) * mb;
            const float r2 = (c1[p] - osample(c2, tf)) * mf;
            w1 |= (uint64_t)(uint8_t)(int8_t)sgn(r1) << (8 * (c - c0));
            w2 |= (uint64_t)(uint8_t)(int8_t)sgn(r2) << (8 * (c - c0));
            lsum += fabsf(r1) + fabsf(r2);
        }
        const int64_t o = ((int64_t)b * gridDim.y + blockIdx.y) * hw + p;
        reinterpret_cast<uint64_t*>(sgn1)[o] = w1;
        reinterpret_cast<uint64_t*>(sgn2)[o] = w2;
    }
    if (loss) {
        __shared__ float red[4];
        const float tot = block_sum_256(lsum, red);
        if (threadIdx.x == 0) atomicAdd(loss, tot);
    }
}


// ------------------------------------------------------------------------------------------------
// Per-pixel reductions over channels, in two deterministic steps so that small planes (8x8 .. 32x32)
// still fill the chip: (1) partial sums over one of S channel slices: block = 64 pixels x 4
// sub-slices, grid (ceil(hw/64), S, B), written to part[b][s][p]; (2) an elementwise kernel adds the S
// partials in a fixed order.  MODE 0: sum x^2 (column norms), MODE 1: sum x*y (<V, dV>), MODE 2: sum (x/n[p])*y
// (<V, dV> with V = X/|X| rebuilt from X: the same quotient normalize wrote, so V need not be stored).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void chan_partial_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            float* __restrict__ part, int C, int hw, int S,
                                                            const float* __restrict__ nrm = nullptr) {
    __shared__ float red[4][64];
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + px;
    const int s = blockIdx.y, b = blockIdx.z;
    const int cper = (C + S - 1) / S;
    const int cbeg = s * cper, cend = min(cbeg + cper, C);
    const int64_t base = (int64_t)b * C * hw;
    float acc = 0.f;
    if (p < hw) {
        const float n = (MODE == 2) ? nrm[(int64_t)b * hw + p] : 1.f;
        for (int c = cbeg + sl; c < cend; c += 4) {
            const int64_t o = base + (int64_t)c * hw + p;
            acc = (MODE == 0) ? fmaf(x[o], x[o], acc) : (MODE == 1 ? fmaf(x[o], y[o], acc) : fmaf(x[o] / n, y[o], acc));
        }
    }
    red[sl][px] = acc;
    __syncthreads();
    if (sl == 0 && p < hw) part[((int64_t)b * S + s) * hw + p] = red[0][px] + red[1][px] + red[2][px] + red[3][px];
}

constexpr int ECPT = 8;  // channels per thread in the elementwise kernels

// vt = x / |x[p]|, nrm[b][p] = |x[p]|.  grid (ceil(hw/256), ceil(C/ECPT), B)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ cs,
                                                         const float* __restrict__ part, float* __restrict__ vt,
                                                         float* __restrict__ nrm, half_t* __restrict__ vh,
                                                         half_t* __restrict__ vl, int C, int hw, int S) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int b = blockIdx.z, c0 = blockIdx.y * ECPT, cend = min(c0 + ECPT, C);
    float ss = 0.f;
    for (int s = 0; s < S; ++s) ss += part[((int64_t)b * S + s) * hw + p];
    const float n = sqrtf(ss);
    if (blockIdx.y == 0) nrm[(int64_t)b * hw + p] = n;
    for (int c = c0; c < cend; ++c) {
        const int64_t o = ((int64_t)b * C + c) * hw + p;
        const float val = cs[o] / n;
        vt[o] = val;
        if (vh) {  // V = vh + vl to ~2^-22 (|V| <= 1): operands of the fp16-MFMA form of S V
            const half_t hi16 = (half_t)val;
            vh[o] = hi16;
            vl[o] = (half_t)(val - (float)hi16);
        }
    }
}

static int chan_slices(int hw, int B, int C) {
    const int blocks = ((hw + 63) / 64) * B;
    int S = (2048 + blocks - 1) / blocks;
    if (S > 32) S = 32;
    if (S > C / 4) S = C / 4 > 0 ? C / 4 : 1;
    return S < 1 ? 1 : S;
}

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM core: block = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 2 x 2
// MFMA 32x32x2 tiles, K chunk 16 staged in LDS as As[k][128], Bs[k][128] (lane-consecutive reads).
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// ------------------------------------------------------------------------------------------------
constexpr int GT = 128;  // block tile
constexpr int GK = 16;   // K chunk

struct GemmAcc {
    floatx16 a[2][2];
};

__device__ __forceinline__ void gemm_zero(GemmAcc& acc) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.a[i][j][r] = 0.f;
}

__device__ __forceinline__ void gemm_chunk(GemmAcc& acc, const float (*As)[GT], const float (*Bs)[GT],
                                           int wm, int wn, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < GK / 2; ++kk) {
        const int k = kk * 2 + hi;
        const float a0 = As[k][wm * 64 + l31], a1 = As[k][wm * 64 + 32 + l31];
        const float b0 = Bs[k][wn * 64 + l31], b1 = Bs[k][wn * 64 + 32 + l31];
        acc.a[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.a[0][0], 0, 0, 0);
        acc.a[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.a[0][1], 0, 0, 0);
        acc.a[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.a[1][0], 0, 0, 0);
        acc.a[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.a[1][1], 0, 0, 0);
    }
}

// loads GK rows x 128 contiguous floats:  dst[i] (i = 0..1) = 4 floats at row (tid/32 + 8*i), col (tid%32)*4
// of the matrix with leading dimension ld whose tile origin is (row0, col0); zero outside [rows, cols).
__device__ __forceinline__ void load_rowmajor_tile(const float* __restrict__ base, int64_t ld, int row0,
                                                   int col0, int rows, int cols, bool vec_ok, int tid,
                                                   floatx4 (&dst)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = row0 + (tid >> 5) + 8 * i;
        const int c = col0 + (tid & 31) * 4;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            const float* p = base + (int64_t)r * ld + c;
            if (vec_ok && c + 3 < cols) {
                v = *reinterpret_cast<const floatx4*>(p);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < cols) v[j] = p[j];
            }
        }
        dst[i] = v;
    }
}

__device__ __forceinline__ void store_rowmajor_tile(float (*S)[GT], int tid, const floatx4 (&src)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        *reinterpret_cast<floatx4*>(&S[(tid >> 5) + 8 * i][(tid & 31) * 4]) = src[i];
}

// This lane's 64 target values in accumulator order, fetched during the K loop so that the strided
// target read (128-byte segments, one tile row apart) overlaps the MFMAs instead of stalling the epilogue
// (measured: 0.4 of 1.2 ms at hw=4096 when read in place).  Half MI=0 is issued at the top of the
// kernel, half MI=1 before the last K chunk, where it reuses the staging registers: 64 more live
// registers would drop the kernel to one block per CU.
struct TilePre {
    float v[2][2][16];
};

template <int MI>
__device__ __forceinline__ void tile_prefetch(TilePre& t, const float* __restrict__ target, int b, int ti, int tj,
                                              int hw, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int mi = MI;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        {
            const int col = tj * GT + wn * 64 + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ti * GT + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                t.v[mi][ni][r] = (row < hw && col < hw) ? target[((int64_t)b * hw + row) * hw + col] : 0.f;
            }
        }
    }
}

// Epilogue shared by the fp32 and the fp16-split Gram kernels: MODE 0 writes sign(G - T) (and, for an
// off-diagonal tile, the transposed tile to the mirror position, staged through `tr` = >= 18 KB of LDS
// that is free once the main loop is done); MODE 1 writes G.

template <int MODE, bool PRE>
__device__ __forceinline__ void gram_epilogue(const GemmAcc& acc, int8_t* tr, const float* __restrict__ target,
                                              int8_t* __restrict__ sgn_out, float* __restrict__ g_out,
                                              float* __restrict__ loss, int b, int ti, int tj, int hw, int tid, const TilePre& pre) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int p0 = ti * GT, q0 = tj * GT;
    const int l31 = lane & 31, hi = lane >> 5;
    const bool mirror = (MODE == 0) && (ti != tj);
    constexpr int TRS = GT + 16;
    float lsum = 0.f;
    int8_t sg[2][2][16];  // this lane's 64 signs (kept for the mirrored tile)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int cl = wn * 64 + ni * 32 + l31;
            const int col = q0 + cl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int row = p0 + rl;
                int8_t v8 = 0;
                if (row < hw && col < hw) {
                    const int64_t o = ((int64_t)b * hw + row) * hw + col;
                    const float gval = acc.a[mi][ni][r];
                    if (MODE == 0) {
                        const float d = gval - (PRE ? pre.v[mi][ni][r] : target[o]);
                        v8 = sign_byte(d);
                        lsum += fabsf(d);
                    } else {
                        g_out[o] = gval;
                    }
                }
                sg[mi][ni][r] = v8;
                if (MODE == 0) tr[rl * TRS + cl] = v8;  // sign tile staged in LDS: rows leave as 16-byte stores
            }
        }
    if (MODE == 0) {
        // write one orientation of the staged tile: LDS row i -> global row (r0 + i), columns c0 ..
        auto flush = [&](int r0, int c0) {
            __syncthreads();
            const bool v16 = (hw % 16 == 0);
            for (int i = tid; i < GT * (GT / 16); i += 256) {
                const int rl = i / (GT / 16), ch = i % (GT / 16);
                const int grow = r0 + rl, gcol = c0 + ch * 16;
                if (grow >= hw) continue;
                int8_t* dst = sgn_out + ((int64_t)b * hw + grow) * hw + gcol;
                const int8_t* src = tr + rl * TRS + ch * 16;
                if (v16 && gcol + 15 < hw) {
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
                } else {
                    for (int e = 0; e < 16; ++e)
                        if (gcol + e < hw) dst[e] = src[e];
                }
            }
        };
        flush(p0, q0);
        if (mirror) {
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int cl = wn * 64 + ni * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        tr[cl * TRS + rl] = sg[mi][ni][r];
                    }
                }
            flush(q0, p0);
        }
    }
    if (MODE == 0 && loss) {
        __shared__ float red[4];
        const float tot = block_sum_256(mirror ? 2.f * lsum : lsum, red);
        if (tid == 0) atomicAdd(loss, tot);
    }
}

__device__ __forceinline__ void tri_tile(int bid, int nt, int& ti, int& tj) {
    ti = 0;
    while (bid >= nt - ti) {
        bid -= nt - ti;
        ++ti;
    }
    tj = ti + bid;
}

// G = V V^T tile.  MODE 0: write int8 sign(G - T) (+ optional loss += sum|G - T|); MODE 1: write G.
// MODE 0 exploits the symmetry of G (bitwise: the same k-ordered fma chain for (p,q) and (q,p)) and of
// the target: only tiles on or above the diagonal are computed -- grid (nt*(nt+1)/2, 1, B) -- and the
// sign tile is also written transposed to its mirror position through LDS, so sv_kernel still reads
// full rows.  Halves the MFMA work and the T stream of the Gram step.  MODE 1: grid (nt, nt, B).
template <int MODE>
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ vt,
                                                    const float* __restrict__ target,
                                                    int8_t* __restrict__ sgn_out, float* __restrict__ g_out,
                                                    float* __restrict__ loss, int C, int hw) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][GT];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][GT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z;
    int ti, tj;
    if (MODE == 0) {
        tri_tile(blockIdx.x, (hw + GT - 1) / GT, ti, tj);
    } else {
        ti = blockIdx.y;
        tj = blockIdx.x;
    }
    const int p0 = ti * GT, q0 = tj * GT;
    const float* v = vt + (int64_t)b * C * hw;
    const bool vec_ok = (hw % 4 == 0);

    GemmAcc acc;
    gemm_zero(acc);
    floatx4 ra[2], rb[2];
    const int nk = (C + GK - 1) / GK;
    load_rowmajor_tile(v, hw, 0, p0, C, hw, vec_ok, tid, ra);
    load_rowmajor_tile(v, hw, 0, q0, C, hw, vec_ok, tid, rb);
    store_rowmajor_tile(As[0], tid, ra);
    store_rowmajor_tile(Bs[0], tid, rb);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) {
            load_rowmajor_tile(v, hw, (kc + 1) * GK, p0, C, hw, vec_ok, tid, ra);
            load_rowmajor_tile(v, hw, (kc + 1) * GK, q0, C, hw, vec_ok, tid, rb);
        }
        gemm_chunk(acc, As[buf], Bs[buf], wm, wn, lane);
        if (kc + 1 < nk) {
            store_rowmajor_tile(As[buf ^ 1], tid, ra);
            store_rowmajor_tile(Bs[buf ^ 1], tid, rb);
        }
        __syncthreads();
    }

    TilePre none;
    gram_epilogue<MODE, false>(acc, reinterpret_cast<int8_t*>(&As[0][0][0]), target, sgn_out, g_out, loss, b, ti, tj,
                               hw, tid, none);
}

// dV^T[c][p] = alpha * sum_q V^T[c][q] * S[q][p]   (S symmetric sign matrix, int8)
// grid (ceil(hw/128), ceil(C/128), B)
__global__ __launch_bounds__(256) void sv_kernel(const float* __restrict__ vt,
                                                  const int8_t* __restrict__ sgn_in,
                                                  float* __restrict__ dvt, int C, int hw, float alpha) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][GT];  // As[k][c]
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][GT];  // Bs[k][p]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * GT, p0 = blockIdx.x * GT;
    const float* v = vt + (int64_t)b * C * hw;
    const int8_t* s = sgn_in + (int64_t)b * hw * hw;
    const bool vec_ok = (hw % 4 == 0);
    const bool svec_ok = (hw % 16 == 0);

    // A loader: thread -> (c = tid/2, 8 k's at (tid%2)*8): two float4 per thread
    // B loader: thread -> (k = tid/16, 8 p's at (tid%16)*8): 8 int8 per thread
    floatx4 ra[2];
    float rb[8];
    auto load_a = [&](int k0) {
        const int c = c0 + (tid >> 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + (tid & 1) * 8 + i * 4;
            floatx4 t = {0.f, 0.f, 0.f, 0.f};
            if (c < C) {
                const float* p = v + (int64_t)c * hw + k;
                if (vec_ok && k + 3 < hw) {
                    t = *reinterpret_cast<const floatx4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < hw) t[j] = p[j];
                }
            }
            ra[i] = t;
        }
    };
    auto load_b = [&](int k0) {
        const int k = k0 + (tid >> 4);
        const int p = p0 + (tid & 15) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = 0.f;
        if (k < hw) {
            const int8_t* sp = s + (int64_t)k * hw + p;
            if (svec_ok && p + 7 < hw) {
                const int2 raw = *reinterpret_cast<const int2*>(sp);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rb[j] = sign_from_byte((uint32_t)raw.x >> (8 * j));
                    rb[4 + j] = sign_from_byte((uint32_t)raw.y >> (8 * j));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (p + j < hw) rb[j] = sign_from_byte((uint8_t)sp[j]);
            }
        }
    };
    auto store_ab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[buf][(tid & 1) * 8 + i * 4 + j][tid >> 1] = ra[i][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[buf][tid >> 4][(tid & 15) * 8 + j] = rb[j];
    };

    GemmAcc acc;
    gemm_zero(acc);
    const int nk = (hw + GK - 1) / GK;
    load_a(0);
    load_b(0);
    store_ab(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) {
            load_a((kc + 1) * GK);
            load_b((kc + 1) * GK);
        }
        gemm_chunk(acc, As[buf], Bs[buf], wm, wn, lane);
        if (kc + 1 < nk) store_ab(buf ^ 1);
        __syncthreads();
    }
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = p0 + wn * 64 + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < C && col < hw) dvt[((int64_t)b * C + row) * hw + col] = acc.a[mi][ni][r] * alpha;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// fp16-split Gram step:  G = Vh Vh^T + Vh Vl^T + Vl Vh^T  (the Vl Vl^T term is < 2^-22) on
// v_mfma_f32_32x32x16_f16 -- 3/16 of the matrix-pipe time of gram_kernel at fp32-class accuracy (every
// product is exact in the fp32 accumulator; |V| <= 1, so Vh + Vl carries V to an absolute 2^-25).
// Both MFMA operands need a pixel's 8 consecutive channels, i.e. V pixel-major: normalize_split_kernel writes
// Vp = Vph + Vpl as (B, hw, C) halfs (64 x 64 tiles transposed through LDS), gram16_kernel stages 128-pixel x
// 32-channel tiles of the four operands in ONE LDS stage (40 KB -> 4 workgroups per CU; the next chunk
// waits in registers), rows of 64 B + 16 B pad (conflict-free ds_read_b128).
// Upper-triangular tiles + mirrored sign tile (gram_epilogue<0>).  Requires C % 8 == 0.
// ------------------------------------------------------------------------------------------------

// Normalisation for the fp16-split GEMMs, one pass over x (same arithmetic and summation order as normalize_kernel): the 64 x 64
// tile of normalised values is written channel-major (vt fp32, vh / vl halfs: operands of S V) straight from the
// registers and pixel-major (vph / vpl: operands of the Gram product) through the LDS transpose.
// grid (ceil(hw/64), ceil(C/64), B), 256 threads
__global__ __launch_bounds__(256) void normalize_split_kernel(const float* __restrict__ cs,
                                                               const float* __restrict__ part, float* __restrict__ vt,
                                                               float* __restrict__ nrm, half_t* __restrict__ vh,
                                                               half_t* __restrict__ vl, half_t* __restrict__ vph,
                                                               half_t* __restrict__ vpl, int C, int hw, int S) {
    __shared__ float tile[64][65];
    __shared__ float nn[64];
    const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    if (threadIdx.x < 64) {
        const int p = p0 + threadIdx.x;
        float ss = 0.f;
        if (p < hw)
            for (int s = 0; s < S; ++s) ss += part[((int64_t)b * S + s) * hw + p];
        const float n = sqrtf(ss);
        nn[threadIdx.x] = n;
        if (blockIdx.y == 0 && p < hw) nrm[(int64_t)b * hw + p] = n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, p = i & 63;
        float val = 0.f;
        if (c0 + c < C && p0 + p < hw) {
            const int64_t o = ((int64_t)b * C + c0 + c) * hw + p0 + p;
            val = cs[o] / nn[p];
            if (vt) vt[o] = val;
            const half_t hi16 = (half_t)val;
            vh[o] = hi16;
            vl[o] = (half_t)(val - (float)hi16);
        }
        tile[c][p] = val;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int p = i >> 6, c = i & 63;
        if (p0 + p < hw && c0 + c < C) {
            const float val = tile[c][p];
            const half_t hi16 = (half_t)val;
            const int64_t o = ((int64_t)b * hw + p0 + p) * C + c0 + c;
            vph[o] = hi16;
            vpl[o] = (half_t)(val - (float)hi16);
        }
    }
}

// K chunk (channels) per staging step, template parameter GK16: 32 (64-byte row segments, 40 KB of LDS: four
// workgroups per CU) for the big planes; 64 for hw <= 1024, where a launch is a few hundred workgroups of 10-20 chunks
// and half as many barrier pairs matter more than occupancy (gram at 8^2 / 16^2 / 32^2: 49 / 55 / 162 -> 44 / 46 / 140 us;
// at 64^2 the 74 KB variant is slower, 737 -> 896 us).  LDS rows of GK16*2 + 16 bytes: an odd multiple of 16.
// grid (nt*(nt+1)/2, 1, B)
template <int GK16>
__global__ __launch_bounds__(256) void gram16_kernel(const half_t* __restrict__ vph, const half_t* __restrict__ vpl,
                                                      const float* __restrict__ target,
                                                      int8_t* __restrict__ sgn_out, float* __restrict__ loss,
                                                      int C, int hw) {
    constexpr int GCPR = GK16 / 8;         // 16-byte chunks per staged row
    constexpr int GNLD = GT * GCPR / 256;  // chunks per thread and array
    constexpr int GROW = GK16 * 2 + 16;    // LDS bytes per tile row
    __shared__ __attribute__((aligned(16))) char lds[4][GT * GROW];  // [Ah, Al, Bh, Bl][pixel row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z;
    int ti, tj;
    tri_tile(blockIdx.x, (hw + GT - 1) / GT, ti, tj);
    const int p0 = ti * GT, q0 = tj * GT;
    const half_t* hb = vph + (int64_t)b * hw * C;
    const half_t* lb = vpl + (int64_t)b * hw * C;
    TilePre pre;
    tile_prefetch<0>(pre, target, b, ti, tj, hw, tid);

    // staging: 4 arrays x 128 rows x 4 chunks of 8 halfs = 2 chunks per thread and array (native vector
    // type: arrays of HIP's uint4 struct are not promoted to registers).  A second register set (two K
    // chunks in flight) was measured: no gain.
    u32x4 rgA[GNLD][4];
    auto load = [&](int k0, u32x4 (&rg)[GNLD][4]) {
#pragma unroll
        for (int i = 0; i < GNLD; ++i) {
            const int ch = tid + i * 256;
            const int srow = ch / GCPR, skc = ch % GCPR;
            const int k = k0 + skc * 8;
            const int pa = p0 + srow, pb = q0 + srow;
            const u32x4 z = {0u, 0u, 0u, 0u};
            const bool ka = k < C;
            rg[i][0] = (ka && pa < hw) ? *reinterpret_cast<const u32x4*>(hb + (int64_t)pa * C + k) : z;
            rg[i][1] = (ka && pa < hw) ? *reinterpret_cast<const u32x4*>(lb + (int64_t)pa * C + k) : z;
            rg[i][2] = (ka && pb < hw) ? *reinterpret_cast<const u32x4*>(hb + (int64_t)pb * C + k) : z;
            rg[i][3] = (ka && pb < hw) ? *reinterpret_cast<const u32x4*>(lb + (int64_t)pb * C + k) : z;
        }
    };
    auto store = [&](const u32x4 (&rg)[GNLD][4]) {
#pragma unroll
        for (int i = 0; i < GNLD; ++i) {
            const int ch = tid + i * 256;
            const int srow = ch / GCPR, skc = ch % GCPR;
#pragma unroll
            for (int a = 0; a < 4; ++a) *reinterpret_cast<u32x4*>(&lds[a][srow * GROW + skc * 16]) = rg[i][a];
        }
    };

    GemmAcc acc;
    gemm_zero(acc);
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < GK16 / 16; ++ks) {
            half8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oa = (wm * 64 + i * 32 + l31) * GROW + ks * 32 + hi * 16;
                const int ob = (wn * 64 + i * 32 + l31) * GROW + ks * 32 + hi * 16;
                ah[i] = *reinterpret_cast<const half8_t*>(&lds[0][oa]);
                al[i] = *reinterpret_cast<const half8_t*>(&lds[1][oa]);
                bh[i] = *reinterpret_cast<const half8_t*>(&lds[2][ob]);
                bl[i] = *reinterpret_cast<const half8_t*>(&lds[3][ob]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc.a[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc.a[i][j], 0, 0, 0);
                    acc.a[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc.a[i][j], 0, 0, 0);
                    acc.a[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc.a[i][j], 0, 0, 0);
                }
        }
    };
    const int nk = (C + GK16 - 1) / GK16;
    load(0, rgA);
    for (int kc = 0; kc + 1 < nk; ++kc) {
        store(rgA);  // the previous chunk's LDS reads are behind the barrier that ended the last iteration
        __syncthreads();
        load((kc + 1) * GK16, rgA);
        compute();
        __syncthreads();
    }
    store(rgA);
    __syncthreads();
    tile_prefetch<1>(pre, target, b, ti, tj, hw, tid);
    compute();
    __syncthreads();
    gram_epilogue<0, true>(acc, reinterpret_cast<int8_t*>(&lds[0][0]), target, sgn_out, (float*)nullptr, loss, b, ti,
                           tj, hw, tid, pre);
}


// ------------------------------------------------------------------------------------------------
// norm backward + Adam, elementwise.  grid (ceil(hw/256), ceil(C/ECPT), B)
//   g = grad_t (if has_t) + (dV - V <V,dV>)/|X| (if has_s);  <V,dV>[b][p] = sum of the S partials
// mode 0: Adam update of cs, m, v;  mode 1: write g to gout (loss_grad entry)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_update_kernel(float* __restrict__ cs, float* __restrict__ m,
                                                           float* __restrict__ v2, TGradArgs tg,
                                                           const float* __restrict__ vt,
                                                           const float* __restrict__ dvt,
                                                           const float* __restrict__ nrm,
                                                           const float* __restrict__ part, float* __restrict__ gout,
                                                           int C, int hw, int S, int has_t, int has_s, int mode,
                                                           AdamArgs a) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int b = blockIdx.z, c0 = blockIdx.y * ECPT, cend = min(c0 + ECPT, C);
    float dot = 0.f, inv_n = 0.f, n = 1.f;
    if (has_s) {
        for (int s = 0; s < S; ++s) dot += part[((int64_t)b * S + s) * hw + p];
        n = nrm[(int64_t)b * hw + p];
        inv_n = 1.f / n;
    }
    // the temporal gradient is formed here from the residual signs (no gradient tensor is written and re-read)
    float tgv[ECPT] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (has_t) {
        TGradPixel tp;
        tp.init(tg, b, p, hw);
        tp.values(tg, blockIdx.y, p, gridDim.y, hw, tgv);
    }
    for (int c = c0; c < cend; ++c) {
        const int64_t o = ((int64_t)b * C + c) * hw + p;
        float g = tgv[c - c0];
        const float x = cs[o];
        if (has_s) g += (dvt[o] - (vt ? vt[o] : x / n) * dot) * inv_n;  // vt == nullptr: V = X/|X| rebuilt (same quotient)
        if (mode == 1) {
            gout[o] = g;
        } else {
            const float mm = a.beta1 * m[o] + (1.f - a.beta1) * g;
            const float vv = a.beta2 * v2[o] + (1.f - a.beta2) * g * g;
            m[o] = mm;
            v2[o] = vv;
            const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
            cs[o] = x - a.step_size * (mm / denom);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t opt_ws_layout(OptWs* w, char* basep, int chunk, int N, int C, int h, int wd, int has_t,
                            int has_s, int n_pairs = 0) {
    // N = frames owned per CFG half; n_pairs = temporal pairs evaluated (N for the single-GPU ring)
    const size_t NP = n_pairs > 0 ? n_pairs : N;
    const size_t B = (size_t)chunk * N, hw = (size_t)h * wd, E = B * C * hw;
    // size query: lay out from a fake non-null base (no memory is touched)
    if (!basep) basep = reinterpret_cast<char*>(static_cast<uintptr_t>(4096));
    char* p = basep;
    OptWs tmp;
    tmp.m = carve<float>(p, E);
    tmp.v = carve<float>(p, E);
    tmp.grad = nullptr;  // (the temporal gradient is formed inside the Adam kernel: no gradient tensor)
    const size_t EP8 = (size_t)chunk * NP * ((C + 7) / 8) * 8 * hw;  // signs: [pair][C/8][hw][8] bytes
    tmp.sgn1 = has_t ? carve<int8_t>(p, EP8) : nullptr;
    tmp.sgn2 = has_t ? carve<int8_t>(p, EP8) : nullptr;
    tmp.rowptr = has_t ? carve<int>(p, (size_t)2 * NP * (hw + 1)) : nullptr;
    tmp.cursor = has_t ? carve<int>(p, (size_t)2 * NP * hw) : nullptr;
    tmp.src = has_t ? carve<int>(p, (size_t)2 * NP * 4 * hw) : nullptr;
    tmp.wgt = has_t ? carve<float>(p, (size_t)2 * NP * 4 * hw) : nullptr;
    tmp.vt = has_s ? carve<float>(p, E) : nullptr;
    tmp.dvt = has_s ? carve<float>(p, E) : nullptr;
    tmp.nrm = has_s ? carve<float>(p, B * hw) : nullptr;
    tmp.part = has_s ? carve<float>(p, B * 32 * hw) : nullptr;
    tmp.dotp = has_s ? carve<float>(p, B * 32 * hw) : nullptr;  // <V, dV> partials per 128-channel tile (opt_fast.hip)
    tmp.vh = has_s ? carve<half_t>(p, E) : nullptr;
    tmp.vl = has_s ? carve<half_t>(p, E) : nullptr;
    tmp.vph = has_s ? carve<half_t>(p, E) : nullptr;
    tmp.vpl = has_s ? carve<half_t>(p, E) : nullptr;
    tmp.ssign = has_s ? carve<int8_t>(p, B * hw * hw) : nullptr;
    tmp.gpart = (has_s && hw <= 64) ? carve<float>(p, B * 8 * 64 * 64) : nullptr;
    if (w) *w = tmp;
    return (size_t)(p - basep);
}

static int opt_check_grid(int chunk, int N, int C) {
    if ((int64_t)chunk * N > 65535 || (C + OCPT - 1) / OCPT > 65535) return FRESCO_EUNSUPPORTED;
    return FRESCO_OK;
}

// one closure evaluation; mode 0 = Adam step, mode 1 = write gradient to gout
// (N = frames owned per CFG half; L describes the temporal layout; Bg = global batch 2*N_total, which
// normalises both loss terms)
static void opt_closure(const OptWs& w, float* cs, const float* fwd_flow, const float* bwd_flow,
                        const float* fwd_occ, const float* bwd_occ, const float* target, int chunk, int N,
                        int C, int h, int wd, float intra_weight, int has_t, int has_s, int mode,
                        float* gout, float* loss, AdamArgs a, hipStream_t st, const TLayout& L, int Bg, int parts = 3) {
    // parts (frame-sharded form): 1 = the launches that read no halo frame (normalise, Gram, S V), 2 = residual signs of
    // every pair + Adam; 3 = both.  The sign launch is independent of part 1's, so 1 then 2 equals 3 bit for bit.
    const int B = chunk * N, hw = h * wd;
    const float kscale = 2.f / ((float)Bg * (float)C * (float)hw);
    const int S = chan_slices(hw, B, C);
    const dim3 egrid((hw + 255) / 256, (C + ECPT - 1) / ECPT, B);
    // S V on fp16 MFMA (V = Vh + Vl) whenever rows are 16-byte aligned; FRESCO_OPT_SV=f32 forces the
    // fp32-MFMA kernel (A/B measurements)
    // (a pure function of the environment, initialised once, thread-safely)
    static const int sv_mode = [] {
        const char* e = getenv("FRESCO_OPT_SV");
        return (e && e[0] == 'f' && e[1] == '3') ? 1 : 0;
    }();
    const bool f16_sv = (hw % 16 == 0) && sv_mode == 0;
    if (has_t && (parts & 2)) {
        dim3 sgrid((hw + 255) / 256, (C + OCPT - 1) / OCPT, chunk * L.n_pairs);
        ProfScope ps(FRESCO_PROF_OPT_TSIGN, B, C, hw, 0, st);
        hipLaunchKernelGGL(temporal_sign_kernel, sgrid, dim3(256), 0, st, cs, bwd_flow, fwd_flow, bwd_occ, fwd_occ,
                           w.sgn1, w.sgn2, loss, L, C, h, wd);
    }
    if (has_s && (parts & 1)) {
        {
            ProfScope ps(FRESCO_PROF_OPT_COLNORM, B, C, hw, 0, st);
            hipLaunchKernelGGL((chan_partial_kernel<0>), dim3((hw + 63) / 64, S, B), dim3(256), 0, st, cs,
                               (const float*)nullptr, w.part, C, hw, S);
            if (f16_sv && C % 8 == 0)  // both GEMMs read the fp16 split: the fp32 V is not stored at all
                hipLaunchKernelGGL(normalize_split_kernel, dim3((hw + 63) / 64, (C + 63) / 64, B), dim3(256), 0, st, cs,
                                   w.part, (float*)nullptr, w.nrm, w.vh, w.vl, w.vph, w.vpl, C, hw, S);
            else
                hipLaunchKernelGGL(normalize_kernel, egrid, dim3(256), 0, st, cs, w.part, w.vt, w.nrm,
                                   f16_sv ? w.vh : (half_t*)nullptr, f16_sv ? w.vl : (half_t*)nullptr, C, hw, S);
        }
        const int nt = (hw + GT - 1) / GT;
        {
            ProfScope ps(FRESCO_PROF_OPT_GRAM, B, C, hw, 0, st);
            if (f16_sv && C % 8 == 0) {
                launch_gram16_plain(w.vph, w.vpl, target, w.ssign, loss ? loss + 1 : nullptr, B, C, hw, st);
            } else
                hipLaunchKernelGGL((gram_kernel<0>), dim3(nt * (nt + 1) / 2, 1, B), dim3(256), 0, st, w.vt, target,
                                   w.ssign, (float*)nullptr, loss ? loss + 1 : nullptr, C, hw);
        }
        const float coef = intra_weight / ((float)Bg * (float)hw * (float)hw);
        {
            ProfScope ps(FRESCO_PROF_OPT_SV, B, C, hw, 0, st);
            if (f16_sv)
                launch_sv16_plain(w.vh, w.vl, w.ssign, w.dvt, nullptr, B, C, hw, 2.f * coef, st);
            else
                hipLaunchKernelGGL(sv_kernel, dim3(nt, (C + GT - 1) / GT, B), dim3(256), 0, st, w.vt, w.ssign,
                                   w.dvt, C, hw, 2.f * coef);
        }
    }
    if (!(parts & 2)) return;
    ProfScope ps(FRESCO_PROF_OPT_ADAM, B, C, hw, 0, st);
    const bool v_stored = !(has_s && (hw % 16 == 0) && sv_mode == 0 && C % 8 == 0);
    if (has_s) {
        if (v_stored)
            hipLaunchKernelGGL((chan_partial_kernel<1>), dim3((hw + 63) / 64, S, B), dim3(256), 0, st, w.vt, w.dvt,
                               w.part, C, hw, S, (const float*)nullptr);
        else
            hipLaunchKernelGGL((chan_partial_kernel<2>), dim3((hw + 63) / 64, S, B), dim3(256), 0, st, cs, w.dvt,
                               w.part, C, hw, S, w.nrm);
    }
    static_assert(OCPT == ECPT, "the temporal gradient is evaluated on the elementwise grid");
    const TGradArgs tg = {w.sgn1, w.sgn2, bwd_occ, fwd_occ, w.rowptr, w.src, w.wgt, L, kscale};
    hipLaunchKernelGGL(adam_update_kernel, egrid, dim3(256), 0, st, cs, w.m, w.v, tg,
                       v_stored ? w.vt : (const float*)nullptr, w.dvt, w.nrm, w.part, gout, C, hw, S, has_t, has_s, mode,
                       a);
}

void launch_gram16_plain(const half_t* vph, const half_t* vpl, const float* target, int8_t* ssign, float* loss, int B,
                         int C, int hw, hipStream_t st) {
    const int nt = (hw + GT - 1) / GT;
    if (hw <= 1024)
        hipLaunchKernelGGL(gram16_kernel<64>, dim3(nt * (nt + 1) / 2, 1, B), dim3(256), 0, st, vph, vpl, target, ssign, loss,
                           C, hw);
    else
        hipLaunchKernelGGL(gram16_kernel<32>, dim3(nt * (nt + 1) / 2, 1, B), dim3(256), 0, st, vph, vpl, target, ssign, loss,
                           C, hw);
}

// loss[0], loss[1] hold raw sums after opt_closure; scale them to the reference's means
__global__ void loss_finalize_kernel(float* loss, float s0, float s1) {
    loss[0] *= s0;
    loss[1] *= s1;
}

}  // namespace fresco

using namespace fresco;

extern "C" size_t fresco_opt_workspace_bytes(int chunk, int N, int C, int h, int w, int has_temporal,
                                             int has_target) {
    if (chunk <= 0 || N <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    return opt_ws_layout(nullptr, nullptr, chunk, N, C, h, w, has_temporal, has_target);
}

static int opt_common_checks(const float* cs, const float* fwd_flow, const float* bwd_flow,
                             const float* fwd_occ, const float* bwd_occ, const float* target,
                             void* workspace, size_t workspace_bytes, int chunk, int N, int C, int h, int w,
                             int* has_t, int* has_s, float intra_weight) {
    if (!cs || !workspace || chunk <= 0 || N <= 0 || C <= 0 || h <= 1 || w <= 1) return FRESCO_EINVAL;
    const bool any_t = fwd_flow || bwd_flow || fwd_occ || bwd_occ;
    const bool all_t = fwd_flow && bwd_flow && fwd_occ && bwd_occ;
    if (any_t && !all_t) return FRESCO_EINVAL;
    *has_t = all_t ? 1 : 0;
    *has_s = (target && intra_weight > 0.f) ? 1 : 0;
    if (!*has_t && !*has_s) return FRESCO_EINVAL;
    if (int rc = opt_check_grid(chunk, N, C)) return rc;
    if (workspace_bytes < opt_ws_layout(nullptr, nullptr, chunk, N, C, h, w, *has_t, *has_s))
        return FRESCO_EWORKSPACE;
    return FRESCO_OK;
}

// CSR of the warp adjoints of `n_pairs` pairs (flows / occs indexed by pair); Bg = global batch
static void opt_prepare(const OptWs& ws, const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                        const float* bwd_occ, int n_pairs, int Bg, int C, int h, int w, int has_t,
                        hipStream_t st) {
    if (!has_t) return;
    const int hw = h * w;
    const float kscale = 2.f / ((float)Bg * (float)C * (float)hw);
    hipLaunchKernelGGL(csr_build_kernel, dim3(n_pairs, 2), dim3(1024), 0, st, bwd_flow, fwd_flow, bwd_occ, fwd_occ,
                       ws.rowptr, ws.cursor, ws.src, ws.wgt, n_pairs, h, w, kscale);
}

static AdamArgs adam_args(int it, float lr, float beta1, float beta2, float eps) {
    AdamArgs a;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.step_size = (float)((double)lr / (1.0 - pow((double)beta1, it)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, it));
    a.eps = eps;
    return a;
}

// ---- the four-launch pipeline of opt_fast.hip: dispatch, CFG-half views, optional two-stream form ------------------
// The slice of the workspace that belongs to CFG halves ck0, ck0 + 1, ...: every per-plane array is plane-major and
// the pair-indexed sign arrays are (half, pair)-major, so a half is one contiguous range of each.  The CSR of the warp
// adjoints is indexed by pair only and shared.
static OptWs ws_half(const OptWs& w, int ck0, int N, int NP, int C, int hw) {
    OptWs o = w;
    const size_t pl = (size_t)ck0 * N, E = pl * C * hw;
    o.m += E;
    o.v += E;
    if (o.vt) {
        o.vt += E;
        o.dvt += E;
        o.nrm += pl * hw;
        o.part += pl * 32 * hw;
        o.dotp += pl * 32 * hw;
        o.vh += E;
        o.vl += E;
        o.vph += E;
        o.vpl += E;
        o.ssign += pl * hw * hw;
        if (o.gpart) o.gpart += pl * 8 * 64 * 64;
    }
    if (o.sgn1) {
        const size_t e8 = (size_t)ck0 * NP * ((C + 7) / 8) * 8 * hw;
        o.sgn1 += e8;
        o.sgn2 += e8;
    }
    return o;
}

// The two CFG halves of a batch are INDEPENDENT problems (the temporal term couples the frames of one half, the Gram
// term is per plane, Adam is elementwise): with chunk == 2 they can run as two pipelines on two streams, the second
// one started half an iteration late, so that the HBM-bound launches of one half (prep, adam) run beside the
// MFMA-bound ones of the other (gram, S V) and the partial last rounds of one kernel are filled by the next.
// FRESCO_OPT_SPLIT = 0: one stream; 1: two streams, same start; 2: second half starts behind the first half's Gram launch;
// 3: the MFMA-bound launches of the two halves strictly alternate (half 1's Gram waits for half 0's S V of the same
// iteration, half 0's next Gram for half 1's S V), the HBM-bound launches float beside them -- without the events two
// free-running pipelines fall back into lockstep within two iterations (profiles/r04_opt_trace_split.txt);
// 4: as 3, and a half's adam (+ next prep) additionally waits for the OTHER half's Gram launch to finish, so that the
// HBM-bound launches run beside the S V launch (the least memory-hungry one), never beside a Gram launch.
struct SideStream {  // = the caller's fresco_ctx: nothing of this is process-wide any more (round 6; SURVEY 8b: "no global state")
    std::atomic_flag busy = ATOMIC_FLAG_INIT;  // one call at a time uses a context; a concurrent caller on the SAME context runs on one stream
    int device = -1;                           // the device its stream / events were created on (first use)
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, mid = nullptr, join = nullptr;
    hipEvent_t sv_done[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};    // [half][iteration parity]
    hipEvent_t gram_done[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [half][iteration parity]
};
static void side_stream_destroy(SideStream& t) {  // call with t.busy held
    auto drop = [](hipEvent_t& e) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    };
    drop(t.fork);
    drop(t.mid);
    drop(t.join);
    for (int i = 0; i < 4; ++i) drop(t.sv_done[i >> 1][i & 1]);
    for (int i = 0; i < 4; ++i) drop(t.gram_done[i >> 1][i & 1]);
    if (t.s) (void)hipStreamDestroy(t.s);
    t.s = nullptr;
}
static bool side_stream_ready(SideStream& t) {  // call with t.busy held
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (t.s) return t.device == dev;  // (a context serves the device it was first used on; elsewhere: one stream)
    t.device = dev;
    if (hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) != hipSuccess) {
        t.s = nullptr;
        return false;
    }
    // every event is needed: a null fork / join would leave half 1 unordered with the caller's stream (cs and the
    // workspace would race silently).  Any failure tears the slot down; the caller then runs the one-stream form.
    bool ok = hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&t.mid, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&t.sv_done[i >> 1][i & 1], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&t.gram_done[i >> 1][i & 1], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        side_stream_destroy(t);
    }
    return ok;
}
static int opt_split_mode(int planes_hw) {
    const char* e = getenv("FRESCO_OPT_SPLIT");  // (read per call: the tests switch it)
    const int env = e ? atoi(e) : -1;
    if (env >= 0) return env;
    return planes_hw >= 2048 ? 1 : 0;  // (8 frames: 16 x 16 planes and up; 2.83 -> 2.70, 9.3 -> 8.3, 29.9 -> 29.8 ms per layer; 8 x 8 planes are launch-bound)
}

static int opt_run_impl(SideStream* ctx, float* cs, const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                        const float* bwd_occ, const float* target, void* workspace, size_t workspace_bytes, int chunk,
                        int N, int C, int h, int w, float intra_weight, int iters, float lr, float beta1, float beta2,
                        float eps, void* stream) {
    int has_t = 0, has_s = 0;
    if (int rc = opt_common_checks(cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, workspace,
                                   workspace_bytes, chunk, N, C, h, w, &has_t, &has_s, intra_weight))
        return rc;
    if (iters < 0) return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    OptWs ws;
    opt_ws_layout(&ws, static_cast<char*>(workspace), chunk, N, C, h, w, has_t, has_s);
    const size_t E = (size_t)chunk * N * C * h * w;
    (void)hipMemsetAsync(ws.m, 0, E * sizeof(float), st);
    (void)hipMemsetAsync(ws.v, 0, E * sizeof(float), st);
    opt_prepare(ws, fwd_flow, bwd_flow, fwd_occ, bwd_occ, N, chunk * N, C, h, w, has_t, st);
    const TLayout L = {N, N, 1, nullptr, nullptr};
    if (opt_fast_ok(C, h, w, has_s)) {
        const int hw = h * w;
        SideStream* sd = (chunk == 2 && !prof_active() && iters > 0) ? ctx : nullptr;  // (no context: one stream)
        int split = sd ? opt_split_mode(N * hw) : 0;
        if (split && sd->busy.test_and_set(std::memory_order_acquire)) split = 0;  // (another host thread owns the side stream)
        struct Release {
            SideStream* p;
            ~Release() {
                if (p) p->busy.clear(std::memory_order_release);
            }
        } release{split ? sd : nullptr};
        if (split && !side_stream_ready(*sd)) split = 0;
        // fork: half 1's stream must see the memsets / CSR build above.  If the record or the wait fails the two-stream form
        // is not entered at all.
        if (split && (hipEventRecord(sd->fork, st) != hipSuccess || hipStreamWaitEvent(sd->s, sd->fork, 0) != hipSuccess)) {
            (void)hipGetLastError();
            split = 0;
        }
        if (!split) {
            opt_fast_begin(ws, cs, chunk * N, C, hw, chunk * N, st);
            for (int it = 1; it <= iters; ++it)
                opt_fast_closure(ws, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, chunk, C, h, w, intra_weight,
                                 has_t, 0, nullptr, nullptr, adam_args(it, lr, beta1, beta2, eps), st, L, chunk * N);
        } else {
            const OptWs w1 = ws_half(ws, 1, N, N, C, hw);
            float* cs1 = cs + (size_t)N * C * hw;
            const float* tg1 = target + (size_t)N * hw * hw;
            opt_fast_begin(ws, cs, N, C, hw, chunk * N, st);  // (fork recorded above: memsets + CSR are behind it)
            if (split == 4) {
                // host issue order: front0(1) front1(1) | back0(1) front0(2) | back1(1) front1(2) | back0(2) front0(3) | ...
                auto issue = [&](int half, int it, int parts) {
                    FastSync y;
                    y.parts = parts;
                    const int par = it & 1;
                    if (parts & 1) {
                        // Gram of half 0 waits for S V of half 1 of the previous iteration, Gram of half 1 for S V of half 0 of this one
                        y.wait_before_gram = half == 0 ? (it > 1 ? sd->sv_done[1][par ^ 1] : nullptr) : sd->sv_done[0][par];
                        y.record_after_gram = sd->gram_done[half][par];
                        y.record_after_sv = sd->sv_done[half][par];
                    }
                    if (parts & 2)  // adam(it) of half 0 runs beside S V(it) of half 1, adam(it) of half 1 beside S V(it + 1) of half 0
                        y.wait_before_adam = half == 0 ? sd->gram_done[1][par] : (it < iters ? sd->gram_done[0][par ^ 1] : nullptr);
                    const AdamArgs a = adam_args(it, lr, beta1, beta2, eps);
                    if (half == 0)
                        opt_fast_closure(ws, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, 1, C, h, w, intra_weight, has_t,
                                         0, nullptr, nullptr, a, st, L, chunk * N, &y);
                    else
                        opt_fast_closure(w1, cs1, fwd_flow, bwd_flow, fwd_occ, bwd_occ, tg1, 1, C, h, w, intra_weight, has_t, 0,
                                         nullptr, nullptr, a, sd->s, L, chunk * N, &y);
                };
                opt_fast_begin(w1, cs1, N, C, hw, chunk * N, sd->s);
                issue(0, 1, 1);
                issue(1, 1, 1);
                for (int it = 1; it <= iters; ++it) {
                    issue(0, it, 2);
                    if (it < iters) issue(0, it + 1, 1);
                    issue(1, it, 2);
                    if (it < iters) issue(1, it + 1, 1);
                }
            } else
            for (int it = 1; it <= iters; ++it) {
                const AdamArgs a = adam_args(it, lr, beta1, beta2, eps);
                const int par = it & 1;
                FastSync y0, y1;
                if (split == 3) {
                    y0.wait_before_gram = it > 1 ? sd->sv_done[1][par ^ 1] : nullptr;
                    y0.record_after_sv = sd->sv_done[0][par];
                    y1.wait_before_gram = sd->sv_done[0][par];
                    y1.record_after_sv = sd->sv_done[1][par];
                }
                if (it == 1 && split == 2) y0.record_after_gram = sd->mid;
                opt_fast_closure(ws, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, 1, C, h, w, intra_weight, has_t, 0,
                                 nullptr, nullptr, a, st, L, chunk * N, &y0);
                if (it == 1 && split == 2) (void)hipStreamWaitEvent(sd->s, sd->mid, 0);  // (recorded by the call above)
                if (it == 1) opt_fast_begin(w1, cs1, N, C, hw, chunk * N, sd->s);
                opt_fast_closure(w1, cs1, fwd_flow, bwd_flow, fwd_occ, bwd_occ, tg1, 1, C, h, w, intra_weight, has_t, 0,
                                 nullptr, nullptr, a, sd->s, L, chunk * N, &y1);
            }
            // join: the caller's stream must not run past half 1.  If the event path fails, fall back to a host-side wait
            // for the side stream (correct, merely slower) and report the launch error
            if (hipEventRecord(sd->join, sd->s) != hipSuccess || hipStreamWaitEvent(st, sd->join, 0) != hipSuccess) {
                (void)hipStreamSynchronize(sd->s);
                return FRESCO_ELAUNCH;
            }
        }
        return check_launch();
    }
    for (int it = 1; it <= iters; ++it)
        opt_closure(ws, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, chunk, N, C, h, w, intra_weight,
                    has_t, has_s, 0, nullptr, nullptr, adam_args(it, lr, beta1, beta2, eps), st, L, chunk * N);
    return check_launch();
}

extern "C" int fresco_opt_run(float* cs, const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                              const float* bwd_occ, const float* target, void* workspace,
                              size_t workspace_bytes, int chunk, int N, int C, int h, int w,
                              float intra_weight, int iters, float lr, float beta1, float beta2, float eps,
                              void* stream) {
    return opt_run_impl(nullptr, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, workspace, workspace_bytes, chunk, N, C, h,
                        w, intra_weight, iters, lr, beta1, beta2, eps, stream);
}

extern "C" int fresco_ctx_create(void** ctx) {
    if (!ctx) return FRESCO_EINVAL;
    *ctx = new (std::nothrow) SideStream();
    return *ctx ? FRESCO_OK : FRESCO_EINVAL;
}

extern "C" int fresco_ctx_destroy(void* ctx) {
    if (!ctx) return FRESCO_OK;
    SideStream* t = static_cast<SideStream*>(ctx);
    if (t->busy.test_and_set(std::memory_order_acquire)) return FRESCO_EINVAL;  // (in use by a call)
    if (t->s) {
        int cur = -1;
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != t->device && t->device >= 0;
        if (sw) (void)hipSetDevice(t->device);
        (void)hipStreamSynchronize(t->s);
        side_stream_destroy(*t);
        if (sw) (void)hipSetDevice(cur);
    }
    delete t;
    return FRESCO_OK;
}

extern "C" int fresco_opt_run_ctx(void* ctx, float* cs, const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                                  const float* bwd_occ, const float* target, void* workspace, size_t workspace_bytes,
                                  int chunk, int N, int C, int h, int w, float intra_weight, int iters, float lr,
                                  float beta1, float beta2, float eps, void* stream) {
    return opt_run_impl(static_cast<SideStream*>(ctx), cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, workspace,
                        workspace_bytes, chunk, N, C, h, w, intra_weight, iters, lr, beta1, beta2, eps, stream);
}

extern "C" int fresco_opt_loss_grad(const float* cs, const float* fwd_flow, const float* bwd_flow,
                                    const float* fwd_occ, const float* bwd_occ, const float* target,
                                    float* grad, float* loss, void* workspace, size_t workspace_bytes,
                                    int chunk, int N, int C, int h, int w, float intra_weight,
                                    void* stream) {
    int has_t = 0, has_s = 0;
    if (!grad) return FRESCO_EINVAL;
    if (int rc = opt_common_checks(cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, workspace,
                                   workspace_bytes, chunk, N, C, h, w, &has_t, &has_s, intra_weight))
        return rc;
    hipStream_t st = as_stream(stream);
    OptWs ws;
    opt_ws_layout(&ws, static_cast<char*>(workspace), chunk, N, C, h, w, has_t, has_s);
    opt_prepare(ws, fwd_flow, bwd_flow, fwd_occ, bwd_occ, N, chunk * N, C, h, w, has_t, st);
    if (loss) (void)hipMemsetAsync(loss, 0, 2 * sizeof(float), st);
    AdamArgs a = {0.f, 0.f, 0.f, 1.f, 0.f};
    const TLayout L = {N, N, 1, nullptr, nullptr};
    if (opt_fast_ok(C, h, w, has_s)) {
        opt_fast_begin(ws, cs, chunk * N, C, h * w, chunk * N, st);
        opt_fast_closure(ws, const_cast<float*>(cs), fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, chunk, C, h, w,
                         intra_weight, has_t, 1, grad, loss, a, st, L, chunk * N);
    } else
        opt_closure(ws, const_cast<float*>(cs), fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, chunk, N, C, h, w,
                    intra_weight, has_t, has_s, 1, grad, loss, a, st, L, chunk * N);
    if (loss) {
        const double B = (double)chunk * N, hw = (double)h * w;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, loss, (float)(2.0 / (B * C * hw)),
                           (float)((double)intra_weight / (B * hw * hw)));
    }
    return check_launch();
}

// ---- frame-sharded form (multi-GPU): begin once per optimize_feature call, then one step per Adam
// iteration with the neighbours' boundary frames exchanged by the host in between -------------------
extern "C" size_t fresco_opt_sharded_workspace_bytes(int chunk, int n_loc, int C, int h, int w, int has_temporal,
                                                     int has_target) {
    if (chunk <= 0 || n_loc <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    return opt_ws_layout(nullptr, nullptr, chunk, n_loc, C, h, w, has_temporal, has_target, n_loc + 1);
}

extern "C" int fresco_opt_sharded_begin(const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                                        const float* bwd_occ, void* workspace, size_t workspace_bytes, int chunk,
                                        int n_loc, int N_total, int C, int h, int w, int has_target,
                                        void* stream) {
    if (!workspace || chunk <= 0 || n_loc <= 0 || N_total < n_loc || C <= 0 || h <= 1 || w <= 1) return FRESCO_EINVAL;
    const bool any_t = fwd_flow || bwd_flow || fwd_occ || bwd_occ;
    const bool all_t = fwd_flow && bwd_flow && fwd_occ && bwd_occ;
    if (any_t && !all_t) return FRESCO_EINVAL;
    const int has_t = all_t ? 1 : 0, has_s = has_target ? 1 : 0;
    if (!has_t && !has_s) return FRESCO_EINVAL;
    if (int rc = opt_check_grid(chunk, n_loc + 1, C)) return rc;
    if (workspace_bytes < opt_ws_layout(nullptr, nullptr, chunk, n_loc, C, h, w, has_t, has_s, n_loc + 1))
        return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    OptWs ws;
    opt_ws_layout(&ws, static_cast<char*>(workspace), chunk, n_loc, C, h, w, has_t, has_s, n_loc + 1);
    const size_t E = (size_t)chunk * n_loc * C * h * w;
    (void)hipMemsetAsync(ws.m, 0, E * sizeof(float), st);
    (void)hipMemsetAsync(ws.v, 0, E * sizeof(float), st);
    opt_prepare(ws, fwd_flow, bwd_flow, fwd_occ, bwd_occ, n_loc + 1, chunk * N_total, C, h, w, has_t, st);
    return check_launch();
}

extern "C" int fresco_opt_sharded_step_part(float* cs, const float* halo_l, const float* halo_r,
                                            const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                                            const float* bwd_occ, const float* target, void* workspace,
                                            size_t workspace_bytes, int chunk, int n_loc, int N_total, int C, int h,
                                            int w, float intra_weight, int it, float lr, float beta1, float beta2,
                                            float eps, int part, void* stream) {
    if (!cs || !workspace || chunk <= 0 || n_loc <= 0 || N_total < n_loc || C <= 0 || h <= 1 || w <= 1 || it < 1)
        return FRESCO_EINVAL;
    if (part < 1 || part > 3) return FRESCO_EINVAL;
    const bool all_t = fwd_flow && bwd_flow && fwd_occ && bwd_occ;
    const int has_t = all_t ? 1 : 0, has_s = (target && intra_weight > 0.f) ? 1 : 0;
    if (!has_t && !has_s) return FRESCO_EINVAL;
    if (has_t && (part & 2) && (!halo_l || !halo_r)) return FRESCO_EINVAL;  // (part 1 reads no halo frame)
    if (workspace_bytes < opt_ws_layout(nullptr, nullptr, chunk, n_loc, C, h, w, has_t, has_s, n_loc + 1))
        return FRESCO_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    OptWs ws;
    opt_ws_layout(&ws, static_cast<char*>(workspace), chunk, n_loc, C, h, w, has_t, has_s, n_loc + 1);
    const TLayout L = {n_loc, n_loc + 1, 0, halo_l, halo_r};
    if (opt_fast_ok(C, h, w, has_s)) {
        if (it == 1 && (part & 1)) opt_fast_begin(ws, cs, chunk * n_loc, C, h * w, chunk * N_total, st);
        FastSync y;
        y.parts = part;
        y.halo_split = part == 3 ? 0 : 1;
        opt_fast_closure(ws, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, chunk, C, h, w, intra_weight, has_t, 0,
                         nullptr, nullptr, adam_args(it, lr, beta1, beta2, eps), st, L, chunk * N_total, &y);
    } else
        opt_closure(ws, cs, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, chunk, n_loc, C, h, w, intra_weight, has_t,
                    has_s, 0, nullptr, nullptr, adam_args(it, lr, beta1, beta2, eps), st, L, chunk * N_total, part);
    return check_launch();
}

extern "C" int fresco_opt_sharded_step(float* cs, const float* halo_l, const float* halo_r,
                                       const float* fwd_flow, const float* bwd_flow, const float* fwd_occ,
                                       const float* bwd_occ, const float* target, void* workspace,
                                       size_t workspace_bytes, int chunk, int n_loc, int N_total, int C, int h,
                                       int w, float intra_weight, int it, float lr, float beta1, float beta2,
                                       float eps, void* stream) {
    return fresco_opt_sharded_step_part(cs, halo_l, halo_r, fwd_flow, bwd_flow, fwd_occ, bwd_occ, target, workspace,
                                        workspace_bytes, chunk, n_loc, N_total, C, h, w, intra_weight, it, lr, beta1, beta2,
                                        eps, 3, stream);
}

extern "C" int fresco_gram_target(const float* x, float* target, void* workspace, size_t workspace_bytes,
                                  int B, int C, int hw, void* stream) {
    if (!x || !target || !workspace || B <= 0 || C <= 0 || hw <= 0) return FRESCO_EINVAL;
    if (B > 65535) return FRESCO_EUNSUPPORTED;
    const size_t E = (size_t)B * C * hw;
    if (workspace_bytes < align_up(E * 4, 256) + align_up((size_t)B * hw * 4, 256) * 33) return FRESCO_EWORKSPACE;
    char* p = static_cast<char*>(workspace);
    float* vt = carve<float>(p, E);
    float* nrm = carve<float>(p, (size_t)B * hw);
    float* part = carve<float>(p, (size_t)B * 32 * hw);
    hipStream_t st = as_stream(stream);
    const int S = chan_slices(hw, B, C);
    if ((C + ECPT - 1) / ECPT > 65535) return FRESCO_EUNSUPPORTED;
    hipLaunchKernelGGL((chan_partial_kernel<0>), dim3((hw + 63) / 64, S, B), dim3(256), 0, st, x,
                       (const float*)nullptr, part, C, hw, S);
    hipLaunchKernelGGL(normalize_kernel, dim3((hw + 255) / 256, (C + ECPT - 1) / ECPT, B), dim3(256), 0, st, x, part,
                       vt, nrm, (half_t*)nullptr, (half_t*)nullptr, C, hw, S);
    const int nt = (hw + GT - 1) / GT;
    hipLaunchKernelGGL((gram_kernel<1>), dim3(nt, nt, B), dim3(256), 0, st, vt, (const float*)nullptr,
                       (int8_t*)nullptr, target, (float*)nullptr, C, hw);
    return check_launch();
}
