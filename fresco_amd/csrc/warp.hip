// Flow-warp family (HBM-bound gathers / elementwise), fp32:
//   flow_warp / bilinear_sample      src/ebsynth/deps/gmflow/gmflow/geometry.py:41-72
//   warp_tensor frame chain          src/flow_utils.py:41-51
//   F.interpolate / max_pool2d prep  src/flow_utils.py:24-35, src/diffusion_hacked.py:437-442
//   Dilate                           src/utils.py:81-93
//   adaptive_instance_normalization  src/utils.py:58-78
//
// Layout is the reference's NCHW.  Sampling taps depend only on (frame, pixel), so one thread owns
// a pixel and a group of CPT channels: the 4 tap indices / weights are computed once and reused,
// consecutive lanes read consecutive pixels of one channel plane (coalesced; the 4-tap gathers of
// a smooth flow stay within a few cache lines of the plane).
#include "common.h"

namespace fresco {

struct Taps {
    int i00, i01, i10, i11;
    float w00, w01, w10, w11;
};

// geometry.py:50-55,65-72: grid = pixel + flow, normalised 2*x/(w-1)-1, grid_sample(align_corners=True)
// maps back with ((g+1)/2)*(size-1); zeros padding -> out-of-range taps get weight 0.
__device__ __forceinline__ Taps make_taps(float fx, float fy, int x, int y, int h, int w) {
    const float gx = 2.f * ((float)x + fx) / (float)(w - 1) - 1.f;
    const float gy = 2.f * ((float)y + fy) / (float)(h - 1) - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f;
    // clamp before the int conversion so that huge / non-finite coordinates stay defined
    const float x0c = fminf(fmaxf(x0f, -2.f), (float)w + 1.f);
    const float y0c = fminf(fmaxf(y0f, -2.f), (float)h + 1.f);
    const int x0 = (int)x0c, y0 = (int)y0c, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < w && x0f == x0c, vx1 = x1 >= 0 && x1 < w && x0f == x0c;
    const bool vy0 = y0 >= 0 && y0 < h && y0f == y0c, vy1 = y1 >= 0 && y1 < h && y0f == y0c;
    const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
    const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
    Taps t;
    t.i00 = cy0 * w + cx0;
    t.i01 = cy0 * w + cx1;
    t.i10 = cy1 * w + cx0;
    t.i11 = cy1 * w + cx1;
    t.w00 = (vx0 && vy0) ? (1.f - tx) * (1.f - ty) : 0.f;
    t.w01 = (vx1 && vy0) ? tx * (1.f - ty) : 0.f;
    t.w10 = (vx0 && vy1) ? (1.f - tx) * ty : 0.f;
    t.w11 = (vx1 && vy1) ? tx * ty : 0.f;
    return t;
}

__device__ __forceinline__ float sample(const float* __restrict__ plane, const Taps& t) {
    return plane[t.i00] * t.w00 + plane[t.i01] * t.w01 + plane[t.i10] * t.w10 + plane[t.i11] * t.w11;
}

constexpr int CPT = 8;  // channels per thread

// grid (ceil(hw/256), ceil(C/CPT), B)
__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ flow,
                                                         float* __restrict__ out, int C, int h, int w,
                                                         int Bf) {
    const int hw = h * w;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const int b = blockIdx.z, c0 = blockIdx.y * CPT;
    const float* fl = flow + (int64_t)(b % Bf) * 2 * hw;
    const Taps t = make_taps(fl[pix], fl[hw + pix], pix % w, pix / w, h, w);
    const int cend = min(c0 + CPT, C);
    for (int c = c0; c < cend; ++c) {
        const int64_t base = ((int64_t)b * C + c) * hw;
        out[base + pix] = sample(x + base, t);
    }
}

// one step of the warp_tensor chain: lat[dst] = lat[dst]*(1-m) + warp(lat[src], flow)*m
// grid (ceil(hw/256), ceil(C/CPT), chunk)
__global__ __launch_bounds__(256) void warp_blend_kernel(float* __restrict__ lat,
                                                          const float* __restrict__ flow,
                                                          const float* __restrict__ occ,
                                                          const float* __restrict__ sal,
                                                          const float* __restrict__ wsal, int src, int dst,
                                                          int N, int C, int h, int w) {
    const int hw = h * w;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const int ck = blockIdx.z, c0 = blockIdx.y * CPT;
    const Taps t = make_taps(flow[pix], flow[hw + pix], pix % w, pix / w, h, w);
    const float m = (1.f - occ[pix]) * sal[pix] * wsal[pix];
    const int cend = min(c0 + CPT, C);
    for (int c = c0; c < cend; ++c) {
        const float* sp = lat + ((int64_t)(ck * N + src) * C + c) * hw;
        float* dp = lat + ((int64_t)(ck * N + dst) * C + c) * hw;
        const float warped = sample(sp, t);
        dp[pix] = dp[pix] * (1.f - m) + warped * m;
    }
}

// F.interpolate(bilinear, align_corners=False): grid (ceil(ho*wo/256), BC)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x,
                                                               float* __restrict__ out, int H, int W,
                                                               int ho, int wo, float rh, float rw,
                                                               float mul) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= ho * wo) return;
    const int oy = o / wo, ox = o % wo;
    const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.f);
    const float sx = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
    const float ly1 = fminf(fmaxf(sy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* p = x + (int64_t)blockIdx.y * H * W;
    const float v00 = p[y0 * W + x0] * mul, v01 = p[y0 * W + x1] * mul;
    const float v10 = p[y1 * W + x0] * mul, v11 = p[y1 * W + x1] * mul;
    out[(int64_t)blockIdx.y * ho * wo + o] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
}

// F.max_pool2d(k): one wave per output element, lanes strided over the k x k window (k is 8..64 on
// this path: a thread-per-output loop would serialise up to 4096 loads)
__global__ __launch_bounds__(256) void max_pool_kernel(const float* __restrict__ x,
                                                        float* __restrict__ out, int H, int W, int k) {
    const int ho = H / k, wo = W / k;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= ho * wo) return;
    const int lane = threadIdx.x & 63;
    const int oy = o / wo, ox = o % wo;
    const float* p = x + (int64_t)blockIdx.y * H * W + (int64_t)oy * k * W + ox * k;
    float m = p[0];
    for (int i = lane; i < k * k; i += 64) m = fmaxf(m, p[(i / k) * W + (i % k)]);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if (lane == 0) out[(int64_t)blockIdx.y * ho * wo + o] = m;
}

// small windows (k < 8): one thread per output element
__global__ __launch_bounds__(256) void max_pool_small_kernel(const float* __restrict__ x,
                                                              float* __restrict__ out, int H, int W, int k) {
    const int ho = H / k, wo = W / k;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= ho * wo) return;
    const int oy = o / wo, ox = o % wo;
    const float* p = x + (int64_t)blockIdx.y * H * W + (int64_t)oy * k * W + ox * k;
    float m = p[0];
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) m = fmaxf(m, p[dy * W + dx]);
    out[(int64_t)blockIdx.y * ho * wo + o] = m;
}

__global__ __launch_bounds__(256) void dilate_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                      int H, int W, int k) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= H * W) return;
    const int y = o / W, xx = o % W, r = (k - 1) / 2;
    const float* p = x + (int64_t)blockIdx.y * H * W;
    float s = 0.f;
    for (int dy = -r; dy <= r; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
        for (int dx = -r; dx <= r; ++dx) s += p[yy * W + min(max(xx + dx, 0), W - 1)];
    }
    out[(int64_t)blockIdx.y * H * W + o] = fminf(fmaxf(s, 0.f), 1.f);
}

// AdaIN: one 256-thread block per (n, c) row
template <typename T>
__global__ __launch_bounds__(256) void adain_kernel(const T* __restrict__ content,
                                                     const T* __restrict__ style, T* __restrict__ out,
                                                     int L, float eps_c, float eps_s) {
    __shared__ float red[4];
    const T* cp = content + (int64_t)blockIdx.x * L;
    const T* sp = style + (int64_t)blockIdx.x * L;
    float sc = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) {
        sc += (float)cp[i];
        ss += (float)sp[i];
    }
    const float mean_c = block_sum_256(sc, red) / (float)L;
    const float mean_s = block_sum_256(ss, red) / (float)L;
    float vc = 0.f, vs = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) {
        const float a = (float)cp[i] - mean_c, b = (float)sp[i] - mean_s;
        vc = fmaf(a, a, vc);
        vs = fmaf(b, b, vs);
    }
    const float var_c = block_sum_256(vc, red) / (float)(L - 1);
    const float var_s = block_sum_256(vs, red) / (float)(L - 1);
    const float std_c = sqrtf(var_c + eps_c), std_s = sqrtf(var_s + eps_s);
    T* op = out + (int64_t)blockIdx.x * L;
    for (int i = threadIdx.x; i < L; i += 256)
        op[i] = (T)((((float)cp[i] - mean_c) / std_c) * std_s + mean_s);
}

// calc_mean_std (src/utils.py:58-67): the two reductions of AdaIN on their own: mean[row], std[row] = sqrt(var_unbiased + eps)
template <typename T>
__global__ __launch_bounds__(256) void chan_mean_std_kernel(const T* __restrict__ x, float* __restrict__ mean,
                                                             float* __restrict__ stdv, int L, float eps) {
    __shared__ float red[4];
    const T* xp = x + (int64_t)blockIdx.x * L;
    float s = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) s += (float)xp[i];
    const float m = block_sum_256(s, red) / (float)L;
    float v = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) {
        const float a = (float)xp[i] - m;
        v = fmaf(a, a, v);
    }
    const float var = block_sum_256(v, red) / (float)(L - 1);
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = m;
        stdv[blockIdx.x] = sqrtf(var + eps);
    }
}

// forward_backward_consistency_check (geometry.py:75-96) + the colour-difference refinement of
// get_flow_and_interframe_paras (diffusion_hacked.py:919-926), one thread per (pair, pixel):
//   occ_f = |fwd + warp(bwd, fwd)| > alpha (|fwd| + |bwd|) + beta   [OR  mean_c |img_n - warp(img_n+1, fwd)| > thr]
//   occ_b = |bwd + warp(fwd, bwd)| > ...                            [OR  mean_c |img_n+1 - warp(img_n, bwd)| > thr]
// Pair n couples frame n with frame (n+1) mod N.  The two tap sets are shared by the flow and the colour
// samples.  grid (ceil(hw/256), N)
__global__ __launch_bounds__(256) void flow_occlusion_kernel(const float* __restrict__ images,
                                                             const float* __restrict__ fwd,
                                                             const float* __restrict__ bwd,
                                                             float* __restrict__ fwd_occ, float* __restrict__ bwd_occ,
                                                             int N, int C, int h, int w, float alpha, float beta,
                                                             float color_thr) {
    const int hw = h * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float* f = fwd + (int64_t)n * 2 * hw;
    const float* b = bwd + (int64_t)n * 2 * hw;
    const float fx = f[p], fy = f[hw + p], bx = b[p], by = b[hw + p];
    const float mag = sqrtf(fx * fx + fy * fy) + sqrtf(bx * bx + by * by);
    const float thr = alpha * mag + beta;
    const Taps tf = make_taps(fx, fy, x, y, h, w);
    const Taps tb = make_taps(bx, by, x, y, h, w);
    const float dfx = fx + sample(b, tf), dfy = fy + sample(b + hw, tf);
    const float dbx = bx + sample(f, tb), dby = by + sample(f + hw, tb);
    bool of = sqrtf(dfx * dfx + dfy * dfy) > thr;
    bool ob = sqrtf(dbx * dbx + dby * dby) > thr;
    if (images) {
        const float* cur = images + (int64_t)n * C * hw;
        const float* nxt = images + (int64_t)((n + 1) % N) * C * hw;
        float sf = 0.f, sb = 0.f;
        for (int c = 0; c < C; ++c) {
            sb += fabsf(nxt[(int64_t)c * hw + p] - sample(cur + (int64_t)c * hw, tb));
            sf += fabsf(cur[(int64_t)c * hw + p] - sample(nxt + (int64_t)c * hw, tf));
        }
        of = of || (sf / (float)C > color_thr);
        ob = ob || (sb / (float)C > color_thr);
    }
    fwd_occ[(int64_t)n * hw + p] = of ? 1.f : 0.f;
    bwd_occ[(int64_t)n * hw + p] = ob ? 1.f : 0.f;
}

}  // namespace fresco

using namespace fresco;

extern "C" int fresco_flow_warp(const float* x, const float* flow, float* out, int B, int C, int h,
                                int w, int Bf, void* stream) {
    if (!x || !flow || !out || B <= 0 || C <= 0 || h <= 0 || w <= 0 || Bf <= 0) return FRESCO_EINVAL;
    if (x == out) return FRESCO_EINVAL;
    dim3 grid((h * w + 255) / 256, (C + CPT - 1) / CPT, B);
    if (grid.y > 65535 || grid.z > 65535) return FRESCO_EUNSUPPORTED;
    hipLaunchKernelGGL(flow_warp_kernel, grid, dim3(256), 0, as_stream(stream), x, flow, out, C, h, w, Bf);
    return check_launch();
}

extern "C" int fresco_resize_bilinear(const float* x, float* out, int BC, int H, int W, int ho, int wo,
                                      float rscale_h, float rscale_w, float mul, void* stream) {
    if (!x || !out || BC <= 0 || H <= 0 || W <= 0 || ho <= 0 || wo <= 0) return FRESCO_EINVAL;
    if (BC > 65535) return FRESCO_EUNSUPPORTED;
    dim3 grid((ho * wo + 255) / 256, BC);
    hipLaunchKernelGGL(resize_bilinear_kernel, grid, dim3(256), 0, as_stream(stream), x, out, H, W, ho, wo,
                       rscale_h, rscale_w, mul);
    return check_launch();
}

extern "C" int fresco_max_pool(const float* x, float* out, int BC, int H, int W, int k, void* stream) {
    if (!x || !out || BC <= 0 || H <= 0 || W <= 0 || k <= 0 || H / k <= 0 || W / k <= 0) return FRESCO_EINVAL;
    if (BC > 65535) return FRESCO_EUNSUPPORTED;
    if (k < 8) {
        dim3 grid(((H / k) * (W / k) + 255) / 256, BC);
        hipLaunchKernelGGL(max_pool_small_kernel, grid, dim3(256), 0, as_stream(stream), x, out, H, W, k);
    } else {
        dim3 grid(((H / k) * (W / k) + 3) / 4, BC);
        hipLaunchKernelGGL(max_pool_kernel, grid, dim3(256), 0, as_stream(stream), x, out, H, W, k);
    }
    return check_launch();
}

extern "C" int fresco_dilate(const float* x, float* out, int BC, int H, int W, int k, void* stream) {
    if (!x || !out || BC <= 0 || H <= 0 || W <= 0 || k <= 0 || (k & 1) == 0) return FRESCO_EINVAL;
    if (BC > 65535) return FRESCO_EUNSUPPORTED;
    dim3 grid((H * W + 255) / 256, BC);
    hipLaunchKernelGGL(dilate_kernel, grid, dim3(256), 0, as_stream(stream), x, out, H, W, k);
    return check_launch();
}

extern "C" int fresco_warp_fuse_chain(float* lat, const float* bwd_flow, const float* fwd_flow,
                                      const float* bwd_occ, const float* fwd_occ, const float* sal,
                                      const float* warp_sal, const float* warp_sal_last, int chunk, int N,
                                      int C, int h, int w, void* stream) {
    if (!lat || !bwd_flow || !fwd_flow || !bwd_occ || !fwd_occ || !sal || !warp_sal || !warp_sal_last)
        return FRESCO_EINVAL;
    if (chunk <= 0 || N <= 0 || C <= 0 || h <= 0 || w <= 0) return FRESCO_EINVAL;
    if (N < 2) return FRESCO_EUNSUPPORTED;  // N == 1 would warp frame 0 into itself in place
    const int hw = h * w;
    dim3 grid((hw + 255) / 256, (C + CPT - 1) / CPT, chunk);
    if (grid.y > 65535 || grid.z > 65535) return FRESCO_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    // flow_utils.py:42-46: frame ii+1 <- blend with warp(frame ii, bwd_flow[ii]); sequential in ii
    for (int ii = 0; ii < N - 1; ++ii)
        hipLaunchKernelGGL(warp_blend_kernel, grid, dim3(256), 0, st, lat, bwd_flow + (int64_t)ii * 2 * hw,
                           bwd_occ + (int64_t)ii * hw, sal + (int64_t)(ii + 1) * hw,
                           warp_sal + (int64_t)ii * hw, ii, ii + 1, N, C, h, w);
    // flow_utils.py:47-51: frame N-1 <- blend with warp(frame 0, fwd_flow[N-1])
    const int ii = N - 1;
    hipLaunchKernelGGL(warp_blend_kernel, grid, dim3(256), 0, st, lat, fwd_flow + (int64_t)ii * 2 * hw,
                       fwd_occ + (int64_t)ii * hw, sal + (int64_t)ii * hw, warp_sal_last, 0, ii, N, C, h, w);
    return check_launch();
}

extern "C" int fresco_adain(const void* content, const void* style, void* out, int rows, int L,
                            float eps_content, float eps_style, int dtype, void* stream) {
    if (!content || !style || !out || rows <= 0 || L <= 1) return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    if (dtype == FRESCO_F16)
        hipLaunchKernelGGL((adain_kernel<half_t>), dim3(rows), dim3(256), 0, st,
                           static_cast<const half_t*>(content), static_cast<const half_t*>(style),
                           static_cast<half_t*>(out), L, eps_content, eps_style);
    else if (dtype == FRESCO_F32)
        hipLaunchKernelGGL((adain_kernel<float>), dim3(rows), dim3(256), 0, st,
                           static_cast<const float*>(content), static_cast<const float*>(style),
                           static_cast<float*>(out), L, eps_content, eps_style);
    else
        return FRESCO_EUNSUPPORTED;
    return check_launch();
}

extern "C" int fresco_chan_mean_std(const void* x, float* mean, float* stdv, int rows, int L, float eps, int dtype,
                                    void* stream) {
    if (!x || !mean || !stdv || rows <= 0 || L <= 1) return FRESCO_EINVAL;
    hipStream_t st = as_stream(stream);
    if (dtype == FRESCO_F16)
        hipLaunchKernelGGL((chan_mean_std_kernel<half_t>), dim3(rows), dim3(256), 0, st, static_cast<const half_t*>(x), mean,
                           stdv, L, eps);
    else if (dtype == FRESCO_F32)
        hipLaunchKernelGGL((chan_mean_std_kernel<float>), dim3(rows), dim3(256), 0, st, static_cast<const float*>(x), mean,
                           stdv, L, eps);
    else
        return FRESCO_EUNSUPPORTED;
    return check_launch();
}

// ------------------------------------------------------------------------------------------------
// (f2) DDPM step pieces of src/pipe_FRESCO.py:14-77, 212-214 as fused elementwise kernels (fp32 math,
// fp16 / fp32 storage).  x0 = (x_t - sqrt(1-abar_t) * eps) / sqrt(abar_t), with eps optionally formed in
// flight by classifier-free guidance  eps = e_u + s (e_c - e_u);  x_{t-1} = c0 x0 + c1 x_t + sigma z.
// ------------------------------------------------------------------------------------------------
namespace fresco {
template <typename T>
__global__ __launch_bounds__(256) void ddpm_x0_kernel(const T* __restrict__ xt, const T* __restrict__ eps_u,
                                                       const T* __restrict__ eps_c, T* __restrict__ x0,
                                                       T* __restrict__ eps_out, int64_t n, float guidance,
                                                       float sqrt_beta, float sqrt_alpha) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float e = (float)eps_u[i];
    if (eps_c) e = e + guidance * ((float)eps_c[i] - e);
    if (eps_out) eps_out[i] = (T)e;
    x0[i] = (T)(((float)xt[i] - sqrt_beta * e) / sqrt_alpha);
}

template <typename T>
__global__ __launch_bounds__(256) void ddpm_prev_kernel(const T* __restrict__ x0, const T* __restrict__ xt,
                                                         const T* __restrict__ noise, T* __restrict__ out,
                                                         int64_t n, int64_t noise_period, float c_x0, float c_xt,
                                                         float sigma) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float mean = c_x0 * (float)x0[i] + c_xt * (float)xt[i];
    out[i] = (T)(mean + sigma * (float)noise[i % noise_period]);
}
}  // namespace fresco

extern "C" int fresco_ddpm_x0(const void* xt, const void* eps_uncond, const void* eps_text, void* x0,
                              void* eps_out, int64_t n, float guidance, float sqrt_beta_prod,
                              float sqrt_alpha_prod, int dtype, void* stream) {
    if (!xt || !eps_uncond || !x0 || n <= 0 || !(sqrt_alpha_prod > 0.f)) return FRESCO_EINVAL;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = as_stream(stream);
    if (dtype == FRESCO_F16)
        hipLaunchKernelGGL((ddpm_x0_kernel<half_t>), grid, dim3(256), 0, st, (const half_t*)xt,
                           (const half_t*)eps_uncond, (const half_t*)eps_text, (half_t*)x0, (half_t*)eps_out, n,
                           guidance, sqrt_beta_prod, sqrt_alpha_prod);
    else if (dtype == FRESCO_F32)
        hipLaunchKernelGGL((ddpm_x0_kernel<float>), grid, dim3(256), 0, st, (const float*)xt,
                           (const float*)eps_uncond, (const float*)eps_text, (float*)x0, (float*)eps_out, n, guidance,
                           sqrt_beta_prod, sqrt_alpha_prod);
    else
        return FRESCO_EUNSUPPORTED;
    return check_launch();
}

extern "C" int fresco_ddpm_prev(const void* x0, const void* xt, const void* noise, void* out, int64_t n,
                                int64_t noise_period, float c_x0, float c_xt, float sigma, int dtype,
                                void* stream) {
    if (!x0 || !xt || !noise || !out || n <= 0 || noise_period <= 0) return FRESCO_EINVAL;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = as_stream(stream);
    if (dtype == FRESCO_F16)
        hipLaunchKernelGGL((ddpm_prev_kernel<half_t>), grid, dim3(256), 0, st, (const half_t*)x0, (const half_t*)xt,
                           (const half_t*)noise, (half_t*)out, n, noise_period, c_x0, c_xt, sigma);
    else if (dtype == FRESCO_F32)
        hipLaunchKernelGGL((ddpm_prev_kernel<float>), grid, dim3(256), 0, st, (const float*)x0, (const float*)xt,
                           (const float*)noise, (float*)out, n, noise_period, c_x0, c_xt, sigma);
    else
        return FRESCO_EUNSUPPORTED;
    return check_launch();
}

extern "C" int fresco_flow_occlusion(const float* images, const float* fwd_flow, const float* bwd_flow,
                                     float* fwd_occ, float* bwd_occ, int N, int C, int H, int W, float alpha,
                                     float beta, float color_thr, void* stream) {
    if (!fwd_flow || !bwd_flow || !fwd_occ || !bwd_occ || N <= 0 || H <= 0 || W <= 0) return FRESCO_EINVAL;
    if (images && C <= 0) return FRESCO_EINVAL;
    if ((int64_t)H * W > (1 << 30) || N > 65535) return FRESCO_EUNSUPPORTED;
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(flow_occlusion_kernel, grid, dim3(256), 0, as_stream(stream), images, fwd_flow, bwd_flow,
                       fwd_occ, bwd_occ, N, C, H, W, alpha, beta, color_thr);
    return check_launch();
}
