"""Register / scratch budgets of the hot kernels, read from the hipcc listing (no GPU needed: hipcc cross-compiles gfx950).

The occupancy each kernel was tuned for is a property of the BUILD, and a refactor can silently lose it (a spill inside
the flash loop cost 2.7x once): two waves per SIMD need <= 256 VGPRs, four need <= 128, and the
SD-1.5 shapes (head dims 40 / 80, K = 320 / 640) must not touch scratch memory."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fresco_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"]
# per-file flags of fresco_amd/csrc/Makefile
EXTRA = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"], "proj.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _listing(src, tmp_path):
    out = str(tmp_path / (src + ".s"))
    subprocess.run([HIPCC] + BASE + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    kernels = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", open(out).read(), re.S):
        blk = m.group(0)
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)  # noqa: E731
        kernels[g("name")] = dict(vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), spill=int(g("vgpr_spill_count")),
                                  scratch=int(g("private_segment_fixed_size")))
    return kernels


def _one(kernels, pattern):
    hits = [(n, k) for n, k in kernels.items() if re.search(pattern, n)]
    assert len(hits) == 1, (pattern, [n for n, _ in hits])
    return hits[0][1]


pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not found")


def test_flash_and_projection_kernels_fit_two_waves_per_simd(tmp_path):
    k = _listing("attn.hip", tmp_path)
    for pat in (r"attn_flash_kernelILi40ELi2E", r"attn_flash_kernelILi80ELi1E"):  # up_blocks.3 / up_blocks.2
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= 256 and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    # the fused K | V projection + pack launch: 5-wave workgroups; at K = 320 two per CU (268 workgroups must not need a
    # second round) -> <= 168 registers; at K = 640 one per CU (93 KB of LDS, 136 workgroups)
    for pat, cap in ((r"kvproj_pack_kernelILi320ELi40E", 168), (r"kvproj_pack_kernelILi640ELi80E", 256)):
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= cap and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    k = _listing("proj.hip", tmp_path)
    for pat in (r"linear_kernelILi320ELi8E", r"linear_kernelILi640ELi8E"):
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= 256 and r["spill"] == 0 and r["scratch"] == 0, (pat, r)


def test_gram_and_sv_kernels_fit_four_waves_per_simd(tmp_path):
    k = _listing("opt_fast.hip", tmp_path)
    # launch_bounds(512, 4): two 8-wave workgroups per CU (the Gram kernel of the run path, both S V tile shapes)
    for pat in (r"gram16y_kernelILb0E", r"gram16y_kernelILb1E", r"sv16b_kernelILi128E", r"sv16b_kernelILi64E"):
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= 128 and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    # the small-plane Gram kernels of round 6: 8-wave / 4-wave workgroups, one per CU by LDS -> two waves per SIMD
    # (gram16sp: one-wave workgroups, one per CU at the batch it is used for -- a whole SIMD's registers are its own)
    for pat, cap in ((r"gram16c_kernelILi8E", 256), (r"gram16c_kernelILi4E", 256), (r"gram16sp_kernelILi8E", 512)):
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= cap and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    for pat in (r"gram16z_kernelILb0E", r"gram16z_kernelILb1E"):  # launch_bounds(256, 3): three 4-wave workgroups per CU
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= 168 and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    for pat in (r"opt_prep_kernel", r"opt_adam_kernelILb0E", r"opt_adam_kernelILb1E"):  # HBM-bound: at least three waves per SIMD, nothing in scratch
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= 168 and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    r = _one(k, r"opt_adam_kernelILb1E")  # the big-plane instantiation (non-temporal m / v / dV): four waves per SIMD
    assert r["vgpr"] + r["agpr"] <= 128, r
    r = _one(_listing("opt.hip", tmp_path), r"adam_update_kernel")
    assert r["spill"] == 0 and r["scratch"] == 0, r


def test_temporal_kernels_do_not_spill(tmp_path):
    k = _listing("temporal.hip", tmp_path)
    for pat in (r"temporal_mfma_kernelILi40ELi1E", r"temporal_mfma_kernelILi80ELi1E", r"temporal_mfma_kernelILi40ELi2E",
                r"temporal_mfma_kernelILi80ELi2E"):
        r = _one(k, pat)
        assert r["spill"] == 0 and r["scratch"] == 0, (pat, r)


def test_flow_network_gemm_fits_two_waves_per_simd_and_sv_reads_stay_split(tmp_path):
    """csrc/flownet.hip's GEMM: 8-wave workgroups, one per CU -> two waves per SIMD: <= 256 VGPRs, nothing in scratch.
    opt_fast.hip's S V kernel: the two sign-row reads of a k-step must stay two ds_read_b64 (32-lane groups, 64 banks: the
    layout its swizzle is conflict-free for); hipcc once fused them into a ds_read2st64_b64 (16-lane groups, 32 banks:
    2-way conflicts, 24 % of the kernel's LDS cycles in profiles/r05_pmc_opt_C640_h64.csv)."""
    k = _listing("flownet.hip", tmp_path)
    for pat in (r"fn_gemm_kernelILi64ELb0E", r"fn_gemm_kernelILi128ELb0E", r"fn_gemm_kernelILi64ELb1E", r"fn_gemm_kernelILi128ELb1E"):  # Lb1: the window-in-LDS convolution form
        r = _one(k, pat)
        assert r["vgpr"] + r["agpr"] <= 256 and r["spill"] == 0 and r["scratch"] == 0, (pat, r)
    _listing("opt_fast.hip", tmp_path)
    txt = open(str(tmp_path / "opt_fast.hip.s")).read()
    i = txt.index("\n_ZN6fresco12sv16b_kernelILi128E")
    body = txt[i:txt.index("s_endpgm", i)]
    assert "ds_read2st64_b64" not in body and body.count("ds_read_b64") >= 4, re.findall(r"ds_read\w+", body)[:20]
