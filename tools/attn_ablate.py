"""Timing ablations of attn_flash_kernel (fresco_amd/csrc/attn.hip) WITHOUT switches in the product source: text edits of a
COPY of attn.hip, one libfresco_hip variant per edit in tools/abl/ (results of the ablated kernels are WRONG: timing only).
    python tools/attn_ablate.py [variant ...]     # build (CPU, cross-compile); time with FRESCO_HIP_LIB=... tools/bench_flash.py
variants: nodma (no pack requests in the common-case step), nobar (counted wait kept, s_barrier dropped, no DMA),
          noreads (fragment ds_reads replaced by a resident fragment), novalu (no exp / cvt)"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fresco_amd", "csrc", "attn.hip")
OUT = os.path.join(ROOT, "tools", "abl")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "
         "-fno-honor-nans -fno-slp-vectorize").split()


def rep(s, old, new, count=1):
    assert old in s, old[:60]
    return s.replace(old, new, count)


def edits(name, s):
    if "nodma" in name or "nobar" in name:
        s = rep(s, "                ring_wait_barrier<NPW_LO>();  // (waves with an extra piece per pack wait for one piece more)\n"
                   "                stage(u + 4, (u + 3) & 3);",
                "                ring_wait_barrier<NPW_LO>();" if "nobar" not in name else
                "                asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\" ::: \"memory\");")
    if "noreads" in name:
        s = rep(s, "vf[kc][db] = *reinterpret_cast<const half8_t*>(vb_ + (kc * 2 * Cfg::DPV + db * 32) * 16);",
                "vf[kc][db] = FAST ? qf[0][0] : *reinterpret_cast<const half8_t*>(vb_ + (kc * 2 * Cfg::DPV + db * 32) * 16);")
        s = rep(s, "        if (!LAST) read_k(kf, slot);",
                "        if (FAST) {\n#pragma unroll\n            for (int ks = 0; ks < Cfg::NKS; ++ks) kf[0][ks] = kf[1][ks] = qf[0][ks];\n        } else if (!LAST) read_k(kf, slot);")
    if "novalu" in name:
        s = rep(s, "                    const float p0 = __builtin_amdgcn_exp2f(s[j][kb][r]);\n"
                   "                    const float p1 = __builtin_amdgcn_exp2f(s[j][kb][r + 1]);",
                "                    const float p0 = FAST ? s[j][kb][r] : __builtin_amdgcn_exp2f(s[j][kb][r]);\n"
                "                    const float p1 = FAST ? s[j][kb][r + 1] : __builtin_amdgcn_exp2f(s[j][kb][r + 1]);")
    return s


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read()
    objs = [os.path.join(ROOT, "fresco_amd", "csrc", "build", o + ".o")
            for o in ("common", "attn32", "proj", "temporal", "warp", "opt", "mapping")]
    for v in sys.argv[1:] or ["base", "nodma", "nobar", "noreads", "novalu", "nodma_noreads", "nobar_noreads_novalu"]:
        s = edits(v, src) if v != "base" else src
        cpp = os.path.join(OUT, "attn_%s.hip" % v)
        open(cpp, "w").write(s.replace('#include "attn_cfg.h"', '#include "../../fresco_amd/csrc/attn_cfg.h"'))
        obj = os.path.join(OUT, "attn_%s.o" % v)
        subprocess.check_call([HIPCC] + FLAGS + ["-c", cpp, "-o", obj])
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o",
                              os.path.join(OUT, "libfresco_hip_%s.so" % v)])
        os.remove(obj)
        print("built", v)


if __name__ == "__main__":
    sys.exit(main())
