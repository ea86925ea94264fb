"""CPU emulation of attn_flash_kernel's arithmetic (fp16 operands, exact fp32 products, P rounded to fp16, row sum from the
rounded P) with the softmax scale either FOLDED into the fp16 Q (one extra rounding of c*q per element) or applied EXACTLY
per score, against an fp64 softmax -- to measure what the fold costs in accuracy as a function of the kernel's own
criterion, the Cauchy-Schwarz logit bound  B = c |q| max|k|  (log2 units; the kernel folds while B <= FOLD_MAX).
Reports, per bound bucket, the worst  |O - ref| / (1e-3 + 1e-3 |ref|)  (the parity bar of tests/: must stay < 1).
    python tools/fold_margin.py
"""
import math
import torch

torch.manual_seed(0)
D, Lq, M = 40, 1024, 4237
LOG2E = 1.4426950408889634


def run(q, k, v, scale, fold):
    c = scale * LOG2E
    qd, kd, vd = q.double(), k.double(), v.double()
    ref = torch.softmax((qd @ kd.T) * scale, -1) @ vd
    if fold:
        qf = (q.float() * c).half().double()            # the kernel: fp32 multiply, one fp16 rounding
        x = (qf @ kd.T).float()                          # MFMA: exact products, fp32 accumulation
    else:
        x = ((qd @ kd.T).float() * c)
    m = x.max(-1, keepdim=True).values.half().float()    # reference point on the fp16 grid
    p = torch.exp2(x - m).half().double()                # P stored as fp16; the ones row sums the ROUNDED P
    o = (p @ vd) / p.sum(-1, keepdim=True)
    err = (o - ref).abs() / (1e-3 + 1e-3 * ref.abs())
    return err.max(dim=1).values                        # per query


def bound(q, k, scale):
    return scale * LOG2E * q.float().norm(dim=1) * k.float().norm(dim=1).max()


def report(name, q, k, v, scale):
    B = bound(q, k, scale)
    ef, ee = run(q, k, v, scale, True), run(q, k, v, scale, False)
    print("%s  (bound %.1f .. %.1f)" % (name, float(B.min()), float(B.max())))
    edges = [0, 8, 12, 16, 20, 24, 32, 48, 1e9]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (B > lo) & (B <= hi)
        if int(sel.sum()):
            print("   B in (%4g, %4g]: %5d queries   worst folded %.3f   worst exact %.3f   (of the 1e-3 bar)"
                  % (lo, hi, int(sel.sum()), float(ef[sel].max()), float(ee[sel].max())))


def sweeps():
    scale = 1 / math.sqrt(D)
    for gain in (0.5, 1.0, 1.5, 2.0, 3.0):
        q = (gain * torch.randn(Lq, D) * torch.linspace(0.5, 1.5, Lq).view(-1, 1)).half()
        k, v = torch.randn(M, D).half(), torch.randn(M, D).half()
        report("N(0,1) k, v; q gain %.1f" % gain, q, k, v, scale)
    # adversarial: every query has a few keys ALIGNED with it (logit = bound for those keys), values far apart
    for a in (1.0, 1.5, 2.0, 3.0):
        q = torch.randn(Lq, D).half()
        k, v = torch.randn(M, D).half(), (3 * torch.randn(M, D)).half()
        idx = torch.randint(0, M, (Lq, 2))
        k[idx[:, 0]] = (a * q.float()).half()
        k[idx[:, 1]] = (a * q.float() * 0.98).half()
        report("aligned keys x%.1f, |v| ~ 3" % a, q, k, v, scale)
    


def logit_range_cases():
    """the inputs of tests/test_gpu_attention.py::test_attention_logit_ranges, kernel decision rule included"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import synth
    for Dh in (40, 80):
        for qgain in (1.0, 2.2, 6.0, 30.0):
            g = synth.gen(int(Dh * 10 + qgain))
            B_, H, Lq_, M_ = 2, 8, 256, 700
            C = H * Dh
            q = (torch.randn(B_, Lq_, C, generator=g) * qgain * torch.linspace(0.3, 1.5, Lq_).view(1, Lq_, 1)).half()
            k = (torch.randn(B_, M_, C, generator=g) * torch.linspace(0.5, 1.6, M_).view(1, M_, 1)).half()
            v = torch.randn(B_, M_, C, generator=g).half()
            sc = 1.0 / math.sqrt(Dh)
            worst = {16.0: 0.0, 24.0: 0.0, 32.0: 0.0}
            for b in range(B_):
                for h in range(H):
                    qq, kk, vv = (t[b, :, h * Dh:(h + 1) * Dh] for t in (q, k, v))
                    Bd = bound(qq, kk, sc)
                    ef, ee = run(qq, kk, vv, sc, True), run(qq, kk, vv, sc, False)
                    for fm in worst:
                        # the kernel folds per WAVE (64 queries at D = 40): all of them must satisfy the criterion
                        rows = 64 if Dh == 40 else 32
                        ok = (Bd <= fm).view(-1, rows).all(1).repeat_interleave(rows)
                        worst[fm] = max(worst[fm], float(torch.where(ok, ef, ee).max()))
            print("logit_ranges D=%d qgain=%4.1f: worst error / bar at FOLD_MAX 16 / 24 / 32: %.3f / %.3f / %.3f"
                  % (Dh, qgain, worst[16.0], worst[24.0], worst[32.0]))


if __name__ == "__main__":
    import sys
    if "ranges" not in sys.argv:
        sweeps()
    logit_range_cases()
