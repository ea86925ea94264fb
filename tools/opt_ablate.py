"""Timing ablations of the feature-optimisation kernels (fresco_amd/csrc/opt_fast.hip) WITHOUT switches in the product
source: text edits of a COPY of opt_fast.hip, one libfresco_hip variant per edit in tools/abl/ (results of the ablated
kernels are WRONG: timing only).
    python tools/opt_ablate.py [variant ...]   # build (CPU, cross-compile)
    FRESCO_HIP_LIB=tools/abl/libfresco_hip_<variant>.so AB_SWITCHES=NONE BENCH_OPT_LAYERS=3 python tools/ab_opt.py 1
Gram (gram16y / gram16z share the epilogue): g_noepi (K loop only), g_nodirect (no LDS staging / stores of the direct
tile), g_nomirror (no mirror stores), g_notgt (no target loads), g_nodma (no operand copies), g_nomfma (no products),
g_nobar (counted waits without s_barrier).  S V: s_nodot (no <V, dV> epilogue), s_nostore (no dV stores), s_nodma,
s_nomfma.  Names combine with '+'."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fresco_amd", "csrc", "opt_fast.hip")
OUT = os.path.join(ROOT, "tools", "abl")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable".split()


def rep(s, old, new, count=1):
    assert s.count(old) >= 1, old[:70]
    return s.replace(old, new, count)


def edits(name, s):
    v = set(name.split("+"))
    if "g_noepi" in v:  # both kernels: the epilogue call and the direct store loop become a dead-code guard on the accumulators
        s = rep(s, "    const float lsum = gram_sign_epilogue<LOSS>(acc, tgt, hw, wgt, tr, wm, wn, l31, hi, sgn_out, b, p0, q0, s_tiled);\n    __syncthreads();\n    for (int idx = tid; idx < 256 * 8; idx += 512) {",
                "    float lsum = 0.f;\n    if (acc[0][0][0] + acc[0][1][3] + acc[1][0][5] + acc[1][1][7] == 12345.f) sgn_out[tid] = 1;\n    for (int idx = tid; idx < 0; idx += 512) {")
    if "g_nodirect" in v:
        s = rep(s, "                tr[(rl + 0) * GX_TRS + cl] = (int8_t)(wv4 & 0xff);\n                tr[(rl + 1) * GX_TRS + cl] = (int8_t)((wv4 >> 8) & 0xff);\n                tr[(rl + 2) * GX_TRS + cl] = (int8_t)((wv4 >> 16) & 0xff);\n                tr[(rl + 3) * GX_TRS + cl] = (int8_t)(wv4 >> 24);",
                "                if (wv4 == 0x12345678u) tr[rl * GX_TRS + cl] = 1;")
        s = rep(s, "    for (int idx = tid; idx < 256 * 8; idx += 512) {\n        const int rl = idx >> 3, ch = idx & 7;\n        const int a = 2 * ti + (rl >> 7);",
                "    for (int idx = tid; idx < 0; idx += 512) {\n        const int rl = idx >> 3, ch = idx & 7;\n        const int a = 2 * ti + (rl >> 7);")
    if "g_nomirror" in v:
        s = rep(s, "            if (wgt == 2) {  // (wave-uniform)", "            if (wgt == 2 && sg[0] == 0x12345678u) {")
    if "g_notgt" in v:
        s = rep(s, "            tnext[r] = wgt ? __builtin_nontemporal_load(tgt + (int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * hw + jj * 32) : 0.f;",
                "            tnext[r] = 0.25f;")
    if "g_nodma" in v:
        s = rep(s, "        GY_PIECE(0, src0);\n        GY_PIECE(1, src1);\n        GY_PIECE(2, src2);", "        (void)m0b; (void)ko;")
    if "g_nomfma" in v:
        old = ("                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[jj], acc[i][jj], 0, 0, 0);\n"
               "                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[jj], acc[i][jj], 0, 0, 0);\n"
               "                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[jj], acc[i][jj], 0, 0, 0);")
        s = rep(s, old, "                    asm volatile(\"\" ::\"v\"(ah[i]), \"v\"(al[i]), \"v\"(bh[jj]), \"v\"(bl[jj]));")
    if "g_nobar" in v:
        s = rep(s, 'asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\\n\\ts_barrier" ::"n"(N_) : "memory");\n}\n\n// ------------------------------------------------------------------------------------------------\n// Epilogue of the 256',
                'asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory");\n}\n\n// ------------------------------------------------------------------------------------------------\n// Epilogue of the 256')
    if "s_nodot" in v:
        s = rep(s, "    if (dotp) {\n        __syncthreads();  // the ring is free\n        const uint32_t mybase", "    if (dotp && alpha == 12345.f) {\n        __syncthreads();\n        const uint32_t mybase")
    if "s_nostore" in v:
        s = rep(s, "                dvt[((int64_t)b * C + c0 + rl) * hw + col] = acc[mi][ni][r];", "                if (acc[mi][ni][r] == 12345.f) dvt[((int64_t)b * C + c0 + rl) * hw + col] = acc[mi][ni][r];")
    if "s_nodma" in v:
        s = rep(s, "        SB_PIECE(o_a, s_a + (int64_t)kc * (SB_TC * SB_VROW));\n        if (CT == 128) SB_PIECE(VARR + wave * 1024, s_l + (int64_t)kc * (SB_TC * SB_VROW));\n        SB_PIECE(2 * VARR + wave * 1024, s_s + (int64_t)kc * SARR);",
                "        (void)m0b;")
    if "s_nomfma" in v:
        s = rep(s, "                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j], acc[i][j], 0, 0, 0);\n                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j], acc[i][j], 0, 0, 0);\n                }\n        }\n        if (kc + 1 < nk) wait_barrier(kc + 2 < nk ? 1 : 0);",
                "                    asm volatile(\"\" ::\"v\"(fa[i][0]), \"v\"(fa[i][1]), \"v\"(fb[j]));\n                }\n        }\n        if (kc + 1 < nk) wait_barrier(kc + 2 < nk ? 1 : 0);")
    # prep (timing of the small planes: which part of the launch is the latency chain)
    if "p_notaps" in v:
        s = rep(s, "                    const float r1 = (q2[p] - osample(q1, tb)) * mb;\n                    const float r2 = (x1[k] - osample(q2, tf)) * mf;",
                "                    const float r1 = (q2[p] - x1[k]) * mb;\n                    const float r2 = (x1[k] - q2[p]) * mf;")
    if "p_nocopies" in v:
        s = rep(s, "            if (do_norm) {\n                half8_t h8, l8;", "            if (do_norm && n == 12345.f) {\n                half8_t h8, l8;")
    if "p_nopart" in v:
        s = rep(s, "            for (int s = 0; s < a.NPB; ++s) ss += a.part[(bn * a.NPB + s) * hw + p];  // same order in every thread",
                "            ss = 1.f + (float)a.NPB;")
    if "p_nosigns" in v:
        s = rep(s, "            if (a.has_t) {\n                const float* c2p = frame_plane(a.cs, L, ck, sb, c0, C, hw);", "            if (a.has_t && n == 12345.f) {\n                const float* c2p = frame_plane(a.cs, L, ck, sb, c0, C, hw);")
    # adam
    if "a_notgrad" in v:
        s = rep(s, "        if (k.has_t) tp.values(k.tg, o, p, C8, hw, tgv);", "        if (k.has_t && n == 12345.f) tp.values(k.tg, o, p, C8, hw, tgv);")
        s = rep(s, "    if (k.has_t) tp.init(k.tg, b, p, hw);", "    if (k.has_t && n == 12345.f) tp.init(k.tg, b, p, hw);")
    if "a_nodot" in v:
        s = rep(s, "        for (int s = 0; s < k.NCT; ++s) dot += k.dotp[((int64_t)b * k.NCT + s) * hw + p];", "        dot = 0.5f;")
    return s


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read()
    objs = [os.path.join(ROOT, "fresco_amd", "csrc", "build", o + ".o")
            for o in ("common", "attn", "attn32", "proj", "temporal", "warp", "opt", "mapping")]
    for v in sys.argv[1:]:
        s = edits(v, src) if v != "base" else src
        cpp = os.path.join(OUT, "opt_fast_%s.hip" % v)
        open(cpp, "w").write(s.replace('#include "opt_shared.h"', '#include "../../fresco_amd/csrc/opt_shared.h"'))
        obj = os.path.join(OUT, "opt_fast_%s.o" % v)
        subprocess.check_call([HIPCC] + FLAGS + ["-c", cpp, "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o",
                              os.path.join(OUT, "libfresco_hip_%s.so" % v)])
        os.remove(obj)
        os.remove(cpp)
        print("built", v)


if __name__ == "__main__":
    sys.exit(main())
