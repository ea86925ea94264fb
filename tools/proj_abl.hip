// Timing ablations of linear_kernel (not part of the product): one binary per -DFRESCO_PROJ_ABL=n
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DFRESCO_PROJ_ABL=n tools/proj_abl.hip \
//         fresco_amd/csrc/common.hip -o build_abl/proj_abl_n
// prints the mean time of the cfg2 up_blocks.3 q,k,v launch (M = 65536, K = 320, 3 x 320 features).
#include "../fresco_amd/csrc/proj.hip"
#include <stdio.h>
#include <vector>

int main() {
    const int M = 65536, K = 320, N = 320;
    std::vector<_Float16> h((size_t)M * K);
    unsigned s = 12345u;
    for (auto& e : h) { s = s * 1664525u + 1013904223u; e = (_Float16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f); }
    _Float16 *x, *w, *o;
    (void)hipMalloc(&x, (size_t)M * K * 2); (void)hipMalloc(&w, (size_t)3 * N * K * 2); (void)hipMalloc(&o, (size_t)3 * M * N * 2);
    (void)hipMemcpy(x, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, h.data(), (size_t)3 * N * K * 2, hipMemcpyHostToDevice);
    fresco_prof_enable(64);
    for (int r = 0; r < 13; ++r)
        fresco_linear(x, K, w, w + (size_t)N * K, w + (size_t)2 * N * K, nullptr, nullptr, nullptr, o, o + (size_t)M * N,
                      o + (size_t)2 * M * N, N, N, N, 3, M, N, K, nullptr);
    (void)hipDeviceSynchronize();
    int tags[64]; int dims[256]; float ms[64];
    const int n = fresco_prof_read(64, tags, dims, ms);
    double tot = 0; int cnt = 0;
    for (int i = 3; i < n; ++i) { tot += ms[i]; ++cnt; }
    printf("PROJ_ABL=%d: q,k,v projection %.1f us\n", FRESCO_PROJ_ABL, 1e3 * tot / cnt);
    return 0;
}
