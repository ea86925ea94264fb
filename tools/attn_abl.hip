// Timing ablations of attn_flash_kernel (not part of the product): one binary per -DFRESCO_ABL=n
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -DFRESCO_ABL=n \
//         tools/attn_abl.hip fresco_amd/csrc/common.hip -o build_abl/abl_n
// prints the mean flash-kernel time of the cfg2 up_blocks.3 spatial-guided launch (HW = 4096, D = 40, 16 key groups).
#include "../fresco_amd/csrc/attn.hip"
#include <stdio.h>
#include <vector>

int main() {
    const int HW = 4096, D = 40, B = 16, H = 8, C = H * D, G = 16, M = 4096;
    const size_t n = (size_t)B * HW * C;
    std::vector<_Float16> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f);
    }
    _Float16 *q, *k, *v, *o;
    void* ws;
    (void)hipMalloc(&q, n * 2); (void)hipMalloc(&k, n * 2); (void)hipMalloc(&v, n * 2); (void)hipMalloc(&o, n * 2);
    (void)hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(v, h.data(), n * 2, hipMemcpyHostToDevice);
    const size_t wsb = fresco_attn_workspace_bytes(G, H, M, D);
    (void)hipMalloc(&ws, wsb);
    fresco_prof_enable(256);
    for (int r = 0; r < 13; ++r) fresco_attn_fwd(q, k, v, nullptr, o, ws, wsb, B, H, HW, D, G, M, HW, 0.0316f, 0.f, nullptr);
    (void)hipDeviceSynchronize();
    int tags[256]; int dims[1024]; float ms[256];
    const int nrec = fresco_prof_read(256, tags, dims, ms);
    double tot = 0; int cnt = 0;
    for (int i = 6; i < nrec; ++i)
        if (tags[i] == FRESCO_PROF_ATTN_FLASH) { tot += ms[i]; ++cnt; }
    printf("ABL=%d: flash %.1f us\n", FRESCO_ABL, 1e3 * tot / cnt);
    return 0;
}
