"""Timing ablations of attn_pipe_kernel (fresco_amd/csrc/attnp.hip) WITHOUT switches in the product source: this script
text-edits a copy of attnp.hip (removing one strand of the pipelined step at a time), builds one libfresco_hip variant per
edit into tools/abl/ (the other objects are the product's), and tools/gpu_attn_ablate.sh times them on a GPU box.
Results of the ablated kernels are WRONG by construction (timing only).
    python tools/attnp_ablate.py [variant ...]   # build the named (default: all) variants (CPU, cross-compile)
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fresco_amd", "csrc", "attnp.hip")
OUT = os.path.join(ROOT, "tools", "abl")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize".split()


def cut(src, start_pat, end_pat, repl=""):
    a = src.index(start_pat)
    b = src.index(end_pat, a) + len(end_pat)
    return src[:a] + repl + src[b:]


def edits(name, s):
    f0 = s.index("    auto fstep = [&]")
    head, body = s[:f0], s[f0:]
    if "nodma" in name:   # no pack requests inside the pipelined step (the counted wait then never waits)
        body = cut(body, "            if constexpr (G % (NM / NPW) == 1", "stage_piece(t + LEAD, G / (NM / NPW));")
    if "novalu" in name:  # no exp / cvt / mul strand
        body = cut(body, "            constexpr int i0 = G * NI / NM", "            });\n")
    if "nolds" in name:   # no fragment reads
        body = cut(body, "            if constexpr (G >= 1 && G < 1 + 2 * NKS) {", "                });\n            }\n")
    if "nobar" in name:   # no ring wait / barrier at the head of the step
        body = body.replace('        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\\n\\ts_barrier" ::"n"((LEAD - 3) * NPW) : "memory");\n        const char* kb_', '        const char* kb_', 1)
        assert "s_barrier" not in body.split("const char* kb_")[0][-300:]
    if "nomfma" in name:  # no matrix strand
        body = cut(body, "            if constexpr (G < NPV) {", "                else mfma_qk(sNew[j][kb], kf[kb][ks], qfa[j][ks]);\n            }\n")
    s = head + body
    if "lead4" in name:   # round-3 first form: 4-slot-deep requests (pack t+4 in step t)
        assert "LEAD = PIPE_LEAD;" in s
        s = s.replace("LEAD = PIPE_LEAD;", "LEAD = 4;")
    if "skew1" in name:   # cvt one unit behind its exps
        assert "constexpr int CVT_SKEW = 3;" in s
        s = s.replace("constexpr int CVT_SKEW = 3;", "constexpr int CVT_SKEW = 1;")
    if "skew5" in name:
        s = s.replace("constexpr int CVT_SKEW = 3;", "constexpr int CVT_SKEW = 5;")
    return s


VARIANTS = sys.argv[1:] or ["base", "nodma", "novalu", "nolds", "nobar_nodma", "nodma_nolds_nobar",
                            "novalu_nodma_nolds_nobar", "nomfma", "nomfma_nodma_nolds_nobar", "lead4", "skew1", "lead4_skew1"]


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read()
    objs = [os.path.join(ROOT, "fresco_amd", "csrc", "build", o + ".o")
            for o in ("common", "attn", "attn32", "proj", "temporal", "warp", "opt", "mapping")]
    for v in VARIANTS:
        s = edits(v, src) if v != "base" else src
        cpp = os.path.join(OUT, "attnp_%s.hip" % v)
        open(cpp, "w").write(s.replace('#include "attn_cfg.h"', '#include "../../fresco_amd/csrc/attn_cfg.h"'))
        obj = os.path.join(OUT, "attnp_%s.o" % v)
        subprocess.check_call([HIPCC] + FLAGS + ["-c", cpp, "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o",
                              os.path.join(OUT, "libfresco_hip_%s.so" % v)])
        os.remove(obj)
        print("built", v)


if __name__ == "__main__":
    sys.exit(main())
