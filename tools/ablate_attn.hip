// Timing ablations of attn_flash_kernel (not part of the product): build one binary per -DFRESCO_ABL=n
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans \
//         -DFRESCO_ABL=n tools/ablate_attn.hip fresco_amd/csrc/common.hip -o /tmp/abl_n
// and run it on the GPU box: prints the mean time of the cfg2 up_blocks.3 / up_blocks.2 cross-frame launches.
#include "../fresco_amd/csrc/attn.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int reps = 20;
    const int shapes[2][3] = {{4096, 40, 4237}, {1024, 80, 1056}};
    for (int si = 0; si < 2; ++si) {
        const int HW = shapes[si][0], D = shapes[si][1], M = shapes[si][2];
        const int B = 16, H = 8, C = H * D, G = 2;
        const size_t n = (size_t)B * HW * C;
        std::vector<_Float16> h(n);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f);
        }
        _Float16 *q, *k, *v, *o;
        void* ws;
        hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2);
        hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
        hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice);
        hipMemcpy(v, h.data(), n * 2, hipMemcpyHostToDevice);
        const size_t wsb = fresco_attn_workspace_bytes(G, H, M, D);
        hipMalloc(&ws, wsb);
        fresco_prof_enable(256);
        for (int r = 0; r < reps + 3; ++r)
            fresco_attn_fwd(q, k, v, nullptr, o, ws, wsb, B, H, HW, D, G, M, 8 * HW, 0.158f, 0.f, nullptr);
        hipDeviceSynchronize();
        int tags[256]; int dims[1024]; float ms[256];
        const int nrec = fresco_prof_read(256, tags, dims, ms);
        double tot = 0; int cnt = 0;
        for (int i = 6; i < nrec; ++i)
            if (tags[i] == FRESCO_PROF_ATTN_FLASH) { tot += ms[i]; ++cnt; }
        printf("ABL=%d HW=%d D=%d M=%d: flash %.1f us\n", FRESCO_ABL, HW, D, M, 1e3 * tot / cnt);
        hipFree(q); hipFree(k); hipFree(v); hipFree(o); hipFree(ws);
    }
    return 0;
}
