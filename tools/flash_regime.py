"""Which per-wave regime of attn_flash_kernel do given activations fall into?  A CPU restatement of the kernel's two
wave-uniform decisions (fresco_amd/csrc/attn.hip, the Cauchy-Schwarz block in front of the key loop), for analysis only:

  folded : scale * log2(e) * |q| * max|k| <= FOLD_MAX (24) for all 64 queries of the wave -> the scale is multiplied
           into the fp16 Q once; otherwise every score is multiplied in fp32 (the "exact" pass, 32 v_pk_mul_f32 per
           64 x 64 block)
  nomax  : after tile 0 has anchored the reference point m, (bound - m) <= NOMAX_THR (14) for all 64 queries -> no
           running-max search on the later tiles

    python tools/flash_regime.py            # the two activation models of bench.py (headline and cfg2c), L3 shapes
"""
import math
import sys

import torch

FOLD_MAX, NOMAX_THR = 24.0, 14.0


def regimes(q, k, scale, rows_per_wave=64):
    """q (Lq, H, D), k (M, H, D) fp16-valued tensors (one batch element / key group).  Returns per-head fractions."""
    q, k = q.float(), k.float()
    c = scale * math.log2(math.e)
    out = []
    for h in range(q.shape[1]):
        qn, kn = q[:, h].norm(dim=1), k[:, h].norm(dim=1)
        kmax = kn.max()
        fold = (c * qn * kmax <= FOLD_MAX).view(-1, rows_per_wave).all(1)           # per wave
        # accumulator units: folded waves carry c in Q, exact waves multiply afterwards; the test is on c * (bound - m)
        m0 = (q[:, h] @ k[:64, h].T).max(dim=1).values * c                           # tile 0 anchors the reference point
        bound = c * qn * kmax * 1.001 + 1e-3
        nomax = (bound - m0 <= NOMAX_THR).view(-1, rows_per_wave).all(1)
        out.append(dict(head=h, bound_max=float((c * qn * kmax).max()), folded=float(fold.float().mean()),
                        nomax=float(nomax.float().mean())))
    return out


def main():
    torch.manual_seed(0)
    HW, M, H, D, C = 4096, 4237, 8, 40, 320
    scale = 1.0 / math.sqrt(D)
    # headline bench: hidden ~ N(0,1), default-init Linear weights (U(-1/sqrt(C), 1/sqrt(C)))
    hidden = torch.randn(HW + M, C).half().float()
    wq, wk = ((torch.rand(C, C) * 2 - 1) / math.sqrt(C) for _ in range(2))
    q = (hidden[:HW] @ wq.T).half().view(HW, H, D)
    k = (hidden[HW:] @ wk.T).half().view(M, H, D)
    for name, (qq, kk) in (("headline (projections of N(0,1) hidden states)", (q, k)),
                           ("cfg2c (q, k ~ N(0,1) per channel)",
                            (torch.randn(HW, H, D).half(), torch.randn(M, H, D).half()))):
        r = regimes(qq, kk, scale)
        print(name)
        print("  logit bound c|q||k| max %.1f (log2 units); waves with folded scale %.0f %%; waves without max search "
              "after tile 0 %.0f %%" % (max(x["bound_max"] for x in r), 100 * sum(x["folded"] for x in r) / len(r),
                                         100 * sum(x["nomax"] for x in r) / len(r)))


if __name__ == "__main__":
    sys.exit(main())
