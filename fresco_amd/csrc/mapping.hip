// FLATTEN pixel correspondences on the GPU: get_single_mapping_ind / get_mapping_ind of
// src/flow_utils.py:56-138 without the per-pixel Python loop (tens of thousands of device syncs per batch
// in the reference).  Integer outputs, bit-exact against the reference's CPU run:
//
//   per frame pair (one 1024-thread block each, mapping_pair_kernel):
//     source f0 -> target f1 = round-half-even(grid + flow), valid iff in range and not occluded;
//     the loop at flow_utils.py:84-97 keeps per target the source with the smallest colour MSE and, among
//     equal errors, the earliest one  ==  lexicographic min of (error bits, source index): one 64-bit
//     atomicMin per valid source;  losers and invalid sources are "unused";  unlinked targets receive the
//     unused sources in ascending order (flow_utils.py:99-101): two block-wide prefix sums + a scatter.
//   chain (mapping_chain_kernel, one block): fwd[i+1] = map_i[fwd[i]], bwd = inverse permutation
//     (torch.sort of a permutation), trajectory mask &= block mask where the link is broken (120-135).
//
// Floating point is kept bit-identical to torch's CPU arithmetic: explicit round-to-nearest mul / add /
// div intrinsics (no fma contraction), error = ((d0^2 + d1^2) + d2^2) / 3.
#include "common.h"

namespace fresco {

// inclusive block scan of one int per thread (1024 threads); returns the inclusive value, total in *total
__device__ __forceinline__ int block_scan_1024(int v, int* buf, int* total) {
    const int tid = threadIdx.x;
    buf[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = tid >= off ? buf[tid - off] : 0;
        __syncthreads();
        buf[tid] += add;
        __syncthreads();
    }
    const int r = buf[tid];
    *total = buf[1023];
    __syncthreads();
    return r;
}

// flow: (P, 2, H, W) already resized, channel 0 = x, 1 = y (not yet divided by scale); occ: (P, H, W)
// resized occlusion (> 0.5 = occluded); frames: (P+1, 3, H*W) resized images.
// outputs per pair: mapping (P, HW) int32, unlinked (P, HW) uint8.  scratch: keys (P, HW) u64, rank (P, 2, HW) int
__global__ __launch_bounds__(1024) void mapping_pair_kernel(const float* __restrict__ flow,
                                                             const float* __restrict__ occ,
                                                             const float* __restrict__ frames,
                                                             int* __restrict__ mapping,
                                                             uint8_t* __restrict__ unlinked,
                                                             unsigned long long* __restrict__ keys,
                                                             int* __restrict__ rank, int H, int W, float scale) {
    __shared__ int buf[1024];
    __shared__ int carry[2];
    const int hw = H * W, i = blockIdx.x, tid = threadIdx.x;
    const float* fl = flow + (int64_t)i * 2 * hw;
    const float* oc = occ + (int64_t)i * hw;
    const float* f0 = frames + (int64_t)i * 3 * hw;        // frames[0] of the pair (the target frame)
    const float* f1 = frames + (int64_t)(i + 1) * 3 * hw;  // frames[1] (the source frame)
    int* map = mapping + (int64_t)i * hw;
    uint8_t* unl = unlinked + (int64_t)i * hw;
    unsigned long long* key = keys + (int64_t)i * hw;
    int* rk_unl = rank + (int64_t)i * 2 * hw;  // rank of a target among the unlinked targets
    int* unused_list = rk_unl + hw;             // unused sources in ascending order
    const unsigned long long EMPTY = ~0ull;

    for (int p = tid; p < hw; p += 1024) key[p] = EMPTY;
    __syncthreads();
    // winner per target: lexicographic min of (error, source)
    for (int s = tid; s < hw; s += 1024) {
        const int y = s / W, x = s % W;
        // flows = interpolate(...)[[1,0]] / scale : channel 1 (y) first  (flow_utils.py:72)
        const float wy = rintf(__fadd_rn((float)y, __fdiv_rn(fl[hw + s], scale)));
        const float wx = rintf(__fadd_rn((float)x, __fdiv_rn(fl[s], scale)));
        const bool ok = wy >= 0.f && wy < (float)H && wx >= 0.f && wx < (float)W && !(oc[s] > 0.5f);
        if (ok) {
            const int t = (int)wy * W + (int)wx;
            const float d0 = __fsub_rn(f1[s], f0[t]);
            const float d1 = __fsub_rn(f1[hw + s], f0[hw + t]);
            const float d2 = __fsub_rn(f1[2 * hw + s], f0[2 * hw + t]);
            const float e = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), 3.f);
            atomicMin(&key[t], ((unsigned long long)__float_as_uint(e) << 32) | (unsigned)s);
        }
    }
    __syncthreads();
    // mapping / unlinked; a source is "used" iff it won its target.  used flags live in rk_unl temporarily.
    for (int p = tid; p < hw; p += 1024) unused_list[p] = 1;  // 1 = unused until proven a winner
    __syncthreads();
    for (int t = tid; t < hw; t += 1024) {
        const unsigned long long k = atomicMin(&key[t], EMPTY);  // atomic read of the L2 value
        if (k != EMPTY) {
            const int s = (int)(k & 0xffffffffu);
            map[t] = s;
            unl[t] = 0;
            unused_list[s] = 0;
        } else {
            map[t] = -1;
            unl[t] = 1;
        }
    }
    __syncthreads();
    // ranks: exclusive prefix sums of the unlinked-target flags and of the unused-source flags
    if (tid < 2) carry[tid] = 0;
    __syncthreads();
    for (int base = 0; base < hw; base += 1024) {
        const int p = base + tid;
        const int fu = p < hw ? (int)unl[p] : 0;
        const int fs = p < hw ? unused_list[p] : 0;
        int tot_u, tot_s;
        const int iu = block_scan_1024(fu, buf, &tot_u);
        const int is = block_scan_1024(fs, buf, &tot_s);
        const int cu = carry[0], cs = carry[1];
        __syncthreads();
        if (p < hw) {
            rk_unl[p] = fu ? cu + iu - 1 : -1;
            // compact the unused sources: stash the destination slot in buf-free memory (key array is done)
            reinterpret_cast<int*>(key)[2 * p] = fs ? cs + is - 1 : -1;
        }
        if (tid == 0) {
            carry[0] = cu + tot_u;
            carry[1] = cs + tot_s;
        }
        __syncthreads();
    }
    for (int p = tid; p < hw; p += 1024) {
        const int slot = reinterpret_cast<int*>(key)[2 * p];
        if (slot >= 0) reinterpret_cast<int*>(key)[2 * slot + 1] = p;  // unused source of that rank
    }
    __syncthreads();
    for (int t = tid; t < hw; t += 1024) {
        const int r = rk_unl[t];
        if (r >= 0) map[t] = reinterpret_cast<int*>(key)[2 * r + 1];
    }
}

// fwd, bwd: (N, HW) int64; mask: (HW, N, N) uint8 (True everywhere on entry is set here)
__global__ __launch_bounds__(1024) void mapping_chain_kernel(const int* __restrict__ mapping,
                                                              const uint8_t* __restrict__ unlinked,
                                                              int64_t* __restrict__ fwd, int64_t* __restrict__ bwd,
                                                              uint8_t* __restrict__ mask, int N, int hw) {
    const int tid = threadIdx.x;
    for (int p = tid; p < hw; p += 1024) {
        fwd[p] = p;
        bwd[p] = p;
        for (int e = 0; e < N * N; ++e) mask[(int64_t)p * N * N + e] = 1;
    }
    __syncthreads();
    for (int i = 0; i + 1 < N; ++i) {
        const int* map = mapping + (int64_t)i * hw;
        const uint8_t* unl = unlinked + (int64_t)i * hw;
        for (int p = tid; p < hw; p += 1024) {
            const int64_t prev = fwd[(int64_t)i * hw + p];
            if (unl[prev]) {  // the trajectory through aligned pixel p breaks between frame i and i+1
                uint8_t* m = mask + (int64_t)p * N * N;
                for (int a = 0; a < N; ++a)
                    for (int b = 0; b < N; ++b)
                        if ((a <= i) != (b <= i)) m[a * N + b] = 0;
            }
            const int64_t nxt = map[prev];
            fwd[(int64_t)(i + 1) * hw + p] = nxt;
            bwd[(int64_t)(i + 1) * hw + nxt] = p;  // fwd[i+1] is a permutation: argsort = inverse
        }
        __syncthreads();
    }
}

}  // namespace fresco

using namespace fresco;

extern "C" size_t fresco_mapping_workspace_bytes(int N, int H, int W) {
    if (N < 1 || H <= 0 || W <= 0) return 0;
    const size_t hw = (size_t)H * W, P = N > 1 ? N - 1 : 1;
    return align_up(P * hw * 4, 256) + align_up(P * hw, 256) + align_up(P * hw * 8, 256) +
           align_up(P * 2 * hw * 4, 256);
}

extern "C" int fresco_mapping_ind(const float* flow, const float* occ, const float* frames, int64_t* fwd_map,
                                  int64_t* bwd_map, uint8_t* mask, void* workspace, size_t workspace_bytes,
                                  int N, int H, int W, float scale, void* stream) {
    if (!flow || !occ || !frames || !fwd_map || !bwd_map || !mask || !workspace) return FRESCO_EINVAL;
    if (N < 1 || H <= 0 || W <= 0 || !(scale > 0.f)) return FRESCO_EINVAL;
    if ((int64_t)H * W > (1 << 24)) return FRESCO_EUNSUPPORTED;
    if (workspace_bytes < fresco_mapping_workspace_bytes(N, H, W)) return FRESCO_EWORKSPACE;
    const size_t hw = (size_t)H * W, P = N > 1 ? N - 1 : 1;
    char* p = static_cast<char*>(workspace);
    int* mapping = carve<int>(p, P * hw);
    uint8_t* unlinked = carve<uint8_t>(p, P * hw);
    unsigned long long* keys = carve<unsigned long long>(p, P * hw);
    int* rank = carve<int>(p, P * 2 * hw);
    hipStream_t st = as_stream(stream);
    if (N > 1)
        hipLaunchKernelGGL(mapping_pair_kernel, dim3(N - 1), dim3(1024), 0, st, flow, occ, frames, mapping, unlinked,
                           keys, rank, H, W, scale);
    hipLaunchKernelGGL(mapping_chain_kernel, dim3(1), dim3(1024), 0, st, mapping, unlinked, fwd_map, bwd_map, mask, N,
                       (int)hw);
    return check_launch();
}
