// Does the request shape of an LDS-DMA piece matter when the data comes from L2?  One workgroup per CU, 8 waves, every wave
// issues global_load_lds_dwordx4 pieces (64 lanes x 16 B = 1 KiB) over a per-workgroup window that stays L2-resident, in three
// lane -> address patterns:  (0) linear: 1 KiB contiguous (8 full 128-byte lines);  (1) 16 rows x 64 B at a row stride of
// 128 B (16 HALF lines: what fn_gemm_kernel's loader asks for -- 32 halfs of one operand plane per row);  (2) 8 rows x 128 B
// at a row stride of 256 B (8 full lines, scattered).  Prints GB/s per pattern.  hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_dma_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, int pattern, int iters, int window, float* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 4096);
    const char* base = src + (size_t)blockIdx.x * window;
    uint32_t off;
    uint32_t step;
    if (pattern == 0) { off = lane * 16; step = 1024; }
    else if (pattern == 1) { off = (lane >> 2) * 128 + (lane & 3) * 16; step = 2048; }
    else if (pattern == 2) { off = (lane >> 3) * 256 + (lane & 7) * 16; step = 2048; }
    else { off = ((lane & 31) >> 2) * 128 + (lane >> 5) * 64 + (lane & 3) * 16; step = 1024; }  // (3) 8 full lines, lanes 0-31 the first halves, 32-63 the second
    uint32_t pos = wave * (window / 8);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const char* p = base + ((pos + off) % (uint32_t)window);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(lds0 + u * 1024) : "memory");
            pos += step;
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem);
}

int main() {
    const int nblk = 256, window = 256 * 1024;  // 64 MB in total: > L2 (32 MB) ... use 96 KB windows for an L2-resident run too
    char* src; float* sink;
    hipMalloc(&src, (size_t)nblk * window);
    hipMemset(src, 1, (size_t)nblk * window);
    hipMalloc(&sink, nblk * sizeof(float));
    hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int win : {96 * 1024, 256 * 1024}) {
        for (int pattern = 0; pattern < 4; ++pattern) {
            const int iters = 2000;
            dma_kernel<<<nblk, 512, 32768>>>(src, pattern, 50, win, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            dma_kernel<<<nblk, 512, 32768>>>(src, pattern, iters, win, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)nblk * 8 * iters * 4 * 1024;
            const double tbs = bytes / (ms * 1e-3) / 1e12;
            printf("window %3d KB per workgroup, pattern %d: %.1f us, %.2f TB/s = %.1f B/clk/CU at 2.1 GHz\n", win / 1024, pattern, ms * 1e3,
                   tbs, tbs * 1e12 / 256 / 2.1e9);
        }
    }
    return 0;
}
