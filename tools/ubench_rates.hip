// Instruction-rate micro-benchmark for gfx950 (not part of the product): cycles per wave-instruction of the
// softmax / MFMA instruction mix of attn_flash_kernel, alone and with a co-resident partner wave on the SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rates.hip -o gpurun_out/ubench_rates && gpurun_out/ubench_rates
// Every test body is inline asm on fixed registers (v0..v127), repeated REP times inside a loop of ITER
// iterations, bracketed by s_memtime; the kernel reports cycles per body execution (min over waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define CLOBBER                                                                                                   \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16",    \
        "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",    \
        "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46",    \
        "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",    \
        "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",    \
        "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91",    \
        "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105",    \
        "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",    \
        "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// 16 independent instructions per body
#define B16(op)                                                                                                  \
    op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7) op(8) op(9) op(10) op(11) op(12) op(13) op(14) op(15)
#define S_(x) #x
#define S(x) S_(x)
#define EXP(i) "v_exp_f32 v" S(i) ", v" S(i) "\n"
#define MUL(i) "v_mul_f32 v" S(i) ", v100, v" S(i) "\n"
#define FMA(i) "v_fma_f32 v" S(i) ", v100, v" S(i) ", v101\n"
#define MAX3(i) "v_max3_f32 v" S(i) ", v100, v" S(i) ", v101\n"
#define CVT(i) "v_cvt_pk_f16_f32 v" S(i) ", v100, v101\n"
#define CVTRTZ(i) "v_cvt_pkrtz_f16_f32 v" S(i) ", v100, v101\n"
#define SWAP(i) "v_permlane16_swap_b32 v" S(i) ", v10" S(i) "\n"
#define PKMUL(i) "v_pk_mul_f32 v[" S(i) "*2:" S(i) "*2+1], v[100:101], v[" S(i) "*2:" S(i) "*2+1]\n"
#define PKADDH(i) "v_pk_add_f16 v" S(i) ", v100, v" S(i) "\n"
#define PKFMAH(i) "v_pk_fma_f16 v" S(i) ", v100, v" S(i) ", v101\n"
#define EXPH(i) "v_exp_f16 v" S(i) ", v" S(i) "\n"
#define MOV(i) "v_mov_b32 v" S(i) ", v100\n"

// MFMA bodies: 4 independent accumulators of 16 regs (32x32) / 8 accumulators of 4 regs (16x16)
#define M32(a) "v_mfma_f32_32x32x16_f16 v[" S(a) ":" S(a) "+15], v[96:99], v[100:103], v[" S(a) ":" S(a) "+15]\n"
#define M16(a) "v_mfma_f32_16x16x32_f16 v[" S(a) ":" S(a) "+3], v[96:99], v[100:103], v[" S(a) ":" S(a) "+3]\n"
#define MF32x4 M32(0) M32(16) M32(32) M32(48)
#define MF16x8 M16(0) M16(4) M16(8) M16(12) M16(16) M16(20) M16(24) M16(28)
// one 32x32 MFMA followed by n independent VALU ops on other registers (64..)
#define EXPb(i) "v_exp_f32 v" S(i) ", v" S(i) "\n"
#define MIX_E(n, a) M32(a) n
#define E1 EXPb(64)
#define E2 EXPb(64) EXPb(65)
#define E3 E2 EXPb(66)
#define E4 E2 EXPb(66) EXPb(67)
#define E5 E4 EXPb(68)
#define E6 E4 EXPb(68) EXPb(69)
#define E8 E4 EXPb(68) EXPb(69) EXPb(70) EXPb(71)
#define MULb(i) "v_mul_f32 v" S(i) ", v100, v" S(i) "\n"
#define U4 MULb(72) MULb(73) MULb(74) MULb(75)
#define U8 U4 MULb(76) MULb(77) MULb(78) MULb(79)

#define DEFK(name, body, nrep)                                                                      \
    __global__ __launch_bounds__(256) void name(unsigned long long* out, int iters) {               \
        unsigned long long t0, t1;                                                                  \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));                              \
        for (int it = 0; it < iters; ++it) {                                                        \
            asm volatile(".rept " #nrep "\n" body ".endr\n" ::: CLOBBER);                            \
        }                                                                                           \
        asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;             \
    }

DEFK(k_exp, B16(EXP), 8)
DEFK(k_mul, B16(MUL), 8)
DEFK(k_fma, B16(FMA), 8)
DEFK(k_max3, B16(MAX3), 8)
DEFK(k_cvt, B16(CVT), 8)
DEFK(k_cvtrtz, B16(CVTRTZ), 8)
DEFK(k_swap, "v_permlane16_swap_b32 v0, v1\n v_permlane16_swap_b32 v2, v3\n v_permlane16_swap_b32 v4, v5\n v_permlane16_swap_b32 v6, v7\n", 8)
DEFK(k_pkmul, "v_pk_mul_f32 v[0:1], v[100:101], v[0:1]\n v_pk_mul_f32 v[2:3], v[100:101], v[2:3]\n v_pk_mul_f32 v[4:5], v[100:101], v[4:5]\n v_pk_mul_f32 v[6:7], v[100:101], v[6:7]\n", 8)
DEFK(k_pkfmah, B16(PKFMAH), 8)
DEFK(k_exph, B16(EXPH), 8)
DEFK(k_mov, B16(MOV), 8)
DEFK(k_m32, MF32x4, 8)
DEFK(k_m32_dep1, M32(0) M32(0) M32(0) M32(0), 8)
DEFK(k_m32_dep2, M32(0) M32(16) M32(0) M32(16), 8)
DEFK(k_m16_dep1, M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0), 8)
DEFK(k_m16_dep2, M16(0) M16(4) M16(0) M16(4) M16(0) M16(4) M16(0) M16(4), 8)
#define DSR(i) "ds_read_b128 v[" S(i) "*4:" S(i) "*4+3], v120 offset:" S(i) "*1024\n"
DEFK(k_dsread, "v_mov_b32 v120, 0\n" DSR(0) DSR(1) DSR(2) DSR(3) DSR(4) DSR(5) DSR(6) DSR(7) "s_waitcnt lgkmcnt(0)\n", 8)
DEFK(k_m32_ds, "v_mov_b32 v120, 0\n" M32(0) DSR(16) M32(16) DSR(17) M32(32) DSR(18) M32(48) DSR(19) "s_waitcnt lgkmcnt(0)\n", 8)
DEFK(k_m16, MF16x8, 8)
DEFK(k_m32_e1, MIX_E(E1, 0) MIX_E(E1, 16) MIX_E(E1, 32) MIX_E(E1, 48), 8)
DEFK(k_m32_e2, MIX_E(E2, 0) MIX_E(E2, 16) MIX_E(E2, 32) MIX_E(E2, 48), 8)
DEFK(k_m32_e3, MIX_E(E3, 0) MIX_E(E3, 16) MIX_E(E3, 32) MIX_E(E3, 48), 8)
DEFK(k_m32_e4, MIX_E(E4, 0) MIX_E(E4, 16) MIX_E(E4, 32) MIX_E(E4, 48), 8)
DEFK(k_m32_e6, MIX_E(E6, 0) MIX_E(E6, 16) MIX_E(E6, 32) MIX_E(E6, 48), 8)
DEFK(k_m32_e8, MIX_E(E8, 0) MIX_E(E8, 16) MIX_E(E8, 32) MIX_E(E8, 48), 8)
DEFK(k_m32_u4, MIX_E(U4, 0) MIX_E(U4, 16) MIX_E(U4, 32) MIX_E(U4, 48), 8)
DEFK(k_m32_u8, MIX_E(U8, 0) MIX_E(U8, 16) MIX_E(U8, 32) MIX_E(U8, 48), 8)
DEFK(k_m32_e4u4, MIX_E(E4 U4, 0) MIX_E(E4 U4, 16) MIX_E(E4 U4, 32) MIX_E(E4 U4, 48), 8)
// 16x16 MFMA + fillers
#define MIX16(n, a) M16(a) n
DEFK(k_m16_e1, MIX16(E1, 0) MIX16(E1, 4) MIX16(E1, 8) MIX16(E1, 12) MIX16(E1, 16) MIX16(E1, 20) MIX16(E1, 24) MIX16(E1, 28), 8)
DEFK(k_m16_e2, MIX16(E2, 0) MIX16(E2, 4) MIX16(E2, 8) MIX16(E2, 12) MIX16(E2, 16) MIX16(E2, 20) MIX16(E2, 24) MIX16(E2, 28), 8)
DEFK(k_m16_u2, MIX16(MULb(72) MULb(73), 0) MIX16(MULb(72) MULb(73), 4) MIX16(MULb(72) MULb(73), 8) MIX16(MULb(72) MULb(73), 12) MIX16(MULb(72) MULb(73), 16) MIX16(MULb(72) MULb(73), 20) MIX16(MULb(72) MULb(73), 24) MIX16(MULb(72) MULb(73), 28), 8)

// heterogeneous pair inside ONE 512-thread workgroup: waves 0-3 (one per SIMD) run body A, waves 4-7 (their
// SIMD partners) run body B; prio = s_setprio level of the B waves
// realistic blocks of the flash kernel (per 64 keys x 64 queries of one wave), tools/ubench_bodies.h:
//  SMX: 64 exponentials on 64 registers, then the 32 packs that depend on them
//  MBL: 28 MFMA 32x32x16 on 8 accumulators (v0..v127), operands static
//  MBD: the same, 14 operand fragments freshly read from LDS in front
#include "ubench_bodies.h"
#define CLOBBER2 CLOBBER, "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v250"

template <int BODY>
__device__ __forceinline__ void run_body(int iters) {
    for (int it = 0; it < iters; ++it) {
        if (BODY == 10) asm volatile(".rept 8\n" BODY_SMX ".endr\n" ::: CLOBBER2);
        if (BODY == 11) asm volatile(".rept 8\n" BODY_MBL ".endr\n" ::: CLOBBER2);
        if (BODY == 12) asm volatile(".rept 8\n" BODY_MBD ".endr\n" ::: CLOBBER2);
        if (BODY == 13) asm volatile(".rept 8\n" BODY_SMX BODY_MBL ".endr\n" ::: CLOBBER2);
        if (BODY == 14) asm volatile(".rept 8\n" BODY_MBL BODY_SMX ".endr\n" ::: CLOBBER2);
        if (BODY == 0) asm volatile(".rept 8\n" MF32x4 ".endr\n" ::: CLOBBER);
        if (BODY == 1) asm volatile(".rept 8\n" B16(MUL) ".endr\n" ::: CLOBBER);
        if (BODY == 2) asm volatile(".rept 8\n" MF16x8 ".endr\n" ::: CLOBBER);
        if (BODY == 3) asm volatile(".rept 8\n" B16(CVT) ".endr\n" ::: CLOBBER);
        if (BODY == 4) asm volatile(".rept 8\n" B16(EXP) ".endr\n" ::: CLOBBER);
        if (BODY == 5) asm volatile(".rept 8\n v_permlane16_swap_b32 v0, v1\n v_permlane16_swap_b32 v2, v3\n v_permlane16_swap_b32 v4, v5\n v_permlane16_swap_b32 v6, v7\n .endr\n" ::: CLOBBER);
        if (BODY == 6) asm volatile(".rept 8\n" B16(FMA) ".endr\n" ::: CLOBBER);
    }
}
template <int BA, int BB>
__global__ __launch_bounds__(512) void k_pair(unsigned long long* out, int iters, int prio) {
    unsigned long long t0, t1;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w >= 4 && prio == 1) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (w < 4) run_body<BA>(iters); else run_body<BB>(iters);
    asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, int);

static void run(const char* name, kern_t k, int n_inst_per_body, unsigned long long* d, unsigned long long* h) {
    const int iters = 200, rep = 8;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int blocks = 256 * bpc;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, blocks * 4 * 8, hipMemcpyDeviceToHost);
        unsigned long long mn = ~0ull, mx = 0; double av = 0;
        for (int i = 0; i < blocks * 4; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; av += h[i]; }
        av /= blocks * 4;
        const double per = av / (iters * rep);
        printf("%-14s waves/SIMD=%d: %7.1f ticks/body (%5.2f per inst, min %.1f max %.1f)  wall %.3f ms -> %.2f ticks/ns\n",
               name, bpc, per, per / n_inst_per_body, (double)mn / (iters * rep), (double)mx / (iters * rep), ms, av / (ms * 1e6));
    }
}

int main() {
    unsigned long long *d, *h;
    hipMalloc(&d, 1024 * 4 * 8); h = (unsigned long long*)malloc(1024 * 4 * 8);
    run("exp_f32", k_exp, 16, d, h);
    run("mul_f32", k_mul, 16, d, h);
    run("fma_f32", k_fma, 16, d, h);
    run("max3_f32", k_max3, 16, d, h);
    run("cvt_pk_f16", k_cvt, 16, d, h);
    run("cvt_pkrtz", k_cvtrtz, 16, d, h);
    run("permlane16sw", k_swap, 4, d, h);
    run("pk_mul_f32", k_pkmul, 4, d, h);
    run("pk_fma_f16", k_pkfmah, 16, d, h);
    run("exp_f16", k_exph, 16, d, h);
    run("mov_b32", k_mov, 16, d, h);
    run("mfma32x4", k_m32, 4, d, h);
    run("mfma16x8", k_m16, 8, d, h);
    run("mfma32 dep1", k_m32_dep1, 4, d, h);
    run("mfma32 dep2", k_m32_dep2, 4, d, h);
    run("mfma16 dep1", k_m16_dep1, 8, d, h);
    run("mfma16 dep2", k_m16_dep2, 8, d, h);
    run("ds_read_b128x8", k_dsread, 8, d, h);
    run("m32+ds_read", k_m32_ds, 4, d, h);
    run("m32+1exp", k_m32_e1, 4, d, h);
    run("m32+2exp", k_m32_e2, 4, d, h);
    run("m32+3exp", k_m32_e3, 4, d, h);
    run("m32+4exp", k_m32_e4, 4, d, h);
    run("m32+6exp", k_m32_e6, 4, d, h);
    run("m32+8exp", k_m32_e8, 4, d, h);
    run("m32+4mul", k_m32_u4, 4, d, h);
    run("m32+8mul", k_m32_u8, 4, d, h);
    run("m32+4exp4mul", k_m32_e4u4, 4, d, h);
    run("m16+1exp", k_m16_e1, 8, d, h);
    run("m16+2exp", k_m16_e2, 8, d, h);
    run("m16+2mul", k_m16_u2, 8, d, h);
    typedef void (*pk_t)(unsigned long long*, int, int);
    struct { const char* name; pk_t k; } pairs[] = {
        {"MFMA32 | exp", k_pair<0, 4>}, {"MFMA32 | mul", k_pair<0, 1>}, {"MFMA16 | exp", k_pair<2, 4>},
        {"MFMA32 | cvt_pk", k_pair<0, 3>}, {"MFMA32 | MFMA32", k_pair<0, 0>}, {"exp | mul", k_pair<4, 1>},
        {"exp | cvt_pk", k_pair<4, 3>}, {"exp | exp", k_pair<4, 4>}, {"exp | swap", k_pair<4, 5>},
        {"cvt_pk | cvt_pk", k_pair<3, 3>}, {"mul | mul", k_pair<1, 1>}, {"fma | fma", k_pair<6, 6>},
        {"cvt_pk | mul", k_pair<3, 1>}, {"swap | swap", k_pair<5, 5>}, {"MFMA32 | swap", k_pair<0, 5>},
        {"SMX | SMX", k_pair<10, 10>}, {"MBL | MBL", k_pair<11, 11>}, {"MBL | SMX", k_pair<11, 10>},
        {"MBD | SMX", k_pair<12, 10>}, {"SM+MB | MB+SM", k_pair<13, 14>}, {"SM+MB | SM+MB", k_pair<13, 13>}};
    for (auto& pr : pairs)
        for (int prio = 0; prio <= 1; ++prio) {
            const int iters = 200;
            hipLaunchKernelGGL(pr.k, dim3(256), dim3(512), 0, 0, d, 10, prio);
            hipLaunchKernelGGL(pr.k, dim3(256), dim3(512), 0, 0, d, iters, prio);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 256 * 8 * 8, hipMemcpyDeviceToHost);
            double a0 = 0, a1 = 0, mn0 = 1e30, mx0 = 0, mn1 = 1e30, mx1 = 0;
            for (int b = 0; b < 256; ++b)
                for (int w = 0; w < 8; ++w) {
                    const double v = (double)h[b * 8 + w] / (iters * 8);
                    if (w < 4) { a0 += v; mn0 = v < mn0 ? v : mn0; mx0 = v > mx0 ? v : mx0; }
                    else { a1 += v; mn1 = v < mn1 ? v : mn1; mx1 = v > mx1 ? v : mx1; }
                }
            printf("pair %-16s prio(B)=%2d: A waves %.1f ticks/body (min %.1f max %.1f), B waves %.1f (min %.1f max %.1f)\n",
                   pr.name, prio, a0 / 1024, mn0, mx0, a1 / 1024, mn1, mx1);
        }
    return 0;
}
