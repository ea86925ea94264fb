// Wave timeline trace of attn_flash_kernel (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -DFRESCO_ATTN_TRACE \
//         tools/attn_trace.hip fresco_amd/csrc/common.hip -o build_abl/attn_trace
// Runs the cfg2 up_blocks.3 spatial-guided launch once and prints, for the waves that shared one SIMD of one CU,
// the s_memtime stamps (relative, in cycles) at the phase boundaries of tiles 8..15:
//   0 tile start | 1 QK issued | 2 V reads + DMA issued | 3 softmax done | 4 barrier passed, K reads issued | 5 PV issued
#include "../fresco_amd/csrc/attn.hip"
#include <stdio.h>
#include <vector>
#include <map>
#include <algorithm>

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
    hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2);
    hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(v, h.data(), n * 2, hipMemcpyHostToDevice);
    const size_t wsb = fresco_attn_workspace_bytes(G, H, M, D);
    hipMalloc(&ws, wsb);
    const int nwg = H * (HW / 256) * B, nw = nwg * 4;
    unsigned int* tr;
    hipMalloc(&tr, (size_t)nw * 64 * 4);
    hipMemset(tr, 0, (size_t)nw * 64 * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(fresco::g_attn_trace), &tr, sizeof(tr));
    for (int r = 0; r < 2; ++r) fresco_attn_fwd(q, k, v, nullptr, o, ws, wsb, B, H, HW, D, G, M, HW, 0.0316f, 0.f, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned int> t((size_t)nw * 64);
    hipMemcpy(t.data(), tr, t.size() * 4, hipMemcpyDeviceToHost);
    // group waves by (xcc, se, sh, cu, simd)
    std::map<unsigned, std::vector<int>> byslot;
    for (int w = 0; w < nw; ++w) {
        const unsigned hw = t[(size_t)w * 64 + 62], xcc = t[(size_t)w * 64 + 63] & 0xf;
        // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] (gfx950: se_id wider)
        const unsigned key = (xcc << 20) | (((hw >> 8) & 0x7f) << 4) | ((hw >> 4) & 3);
        byslot[key].push_back(w);
    }
    printf("%zu distinct (xcc, se/sh/cu, simd) slots for %d waves\n", byslot.size(), nw);
    int shown = 0;
    for (auto& kv : byslot) {
        if (kv.second.size() < 8) continue;
        // sort the waves of this SIMD by their first stamp
        auto& ws_ = kv.second;
        std::sort(ws_.begin(), ws_.end(), [&](int a, int b) { return t[(size_t)a * 64] < t[(size_t)b * 64]; });
        const unsigned t0 = t[(size_t)ws_[0] * 64];
        printf("slot %06x: %zu waves over the launch\n", kv.first, ws_.size());
        for (size_t i = 0; i < ws_.size() && i < 2; ++i) {
            const int w = ws_[i];
            printf("  wave %5d (wg %4d):", w, w / 4);
            for (int tl = 0; tl < 4; ++tl) {
                printf(" |");
                for (int p = 0; p < 6; ++p) printf(" %6u", t[(size_t)w * 64 + tl * 6 + p] - t0);
            }
            printf("\n");
        }
        if (++shown == 4) break;
    }
    return 0;
}
