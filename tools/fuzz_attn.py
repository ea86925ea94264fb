"""Randomised stress of fresco_attn_fwd against an fp32 softmax on the same GPU: shapes, head dims, key groups,
row selections, logit magnitudes (all three kernel paths: folded / exact scale, with / without the max search),
diagonal bias.  usage: python tools/fuzz_attn.py [cases] [seed] [big]
("big": 8 heads and up to 3000 query rows, i.e. more query blocks than CUs -- for kernel variants that walk several
blocks per workgroup)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fresco_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"
g = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
worst = 0.0
for it in range(cases):
    D = [8, 16, 32, 40, 64, 80, 96, 128][ri(0, 7)]
    H = 8 if big else [1, 2, 4, 8][ri(0, 3)]
    C = H * D
    G = ri(1, 3)
    per = ri(2, 3) if big else ri(1, 3)
    B = G * per
    Lq = ri(1500, 3000) if big else ri(1, 700)
    rows_per_group = ri(1, 900)
    use_rows = ri(0, 1) == 1
    gain = [0.3, 1.0, 2.5, 6.0, 20.0][ri(0, 4)]
    bias = [0.0, 0.0, 1.5, -2.0][ri(0, 3)]
    q = (torch.randn(B, Lq, C, generator=g) * gain).half().cuda()
    k = torch.randn(G * rows_per_group, C, generator=g).half().cuda()
    v = torch.randn(G * rows_per_group, C, generator=g).half().cuda()
    scale = 1.0 / math.sqrt(D)
    if use_rows:
        M = ri(1, rows_per_group)
        rows = torch.randperm(rows_per_group, generator=g)[:M].sort().values.to(torch.int32).cuda()
    else:
        M, rows = rows_per_group, None
    if bias != 0.0 and (use_rows or M != Lq):
        bias = 0.0  # the diagonal bias is defined for the square, unselected case (spatial pass)
    out = ops.attention(q, k, v, H, scale, kv_rows=rows, n_groups=G, M=M, group_rows=rows_per_group, diag_bias=bias)
    # reference: fp32 on the GPU
    kk = k.float().view(G, rows_per_group, C)
    vv = v.float().view(G, rows_per_group, C)
    if rows is not None:
        kk, vv = kk[:, rows.long()], vv[:, rows.long()]
    qh = q.float().view(G, per, Lq, H, D).permute(0, 1, 3, 2, 4)
    kh = kk.view(G, 1, M, H, D).permute(0, 1, 3, 2, 4)
    vh = vv.view(G, 1, M, H, D).permute(0, 1, 3, 2, 4)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if bias != 0.0:
        s = s + bias * torch.eye(Lq, M, device="cuda")
    ref = (torch.softmax(s, -1) @ vh).permute(0, 1, 3, 2, 4).reshape(B, Lq, C)
    err = (out.float() - ref).abs()
    tol = 3e-3 + 3e-3 * ref.abs()
    bad = int((err > tol).sum())
    worst = max(worst, float(err.max()))
    ok = bad == 0 and bool(torch.isfinite(out).all())
    print("%3d D=%3d H=%d G=%d per=%d Lq=%3d M=%3d rows=%d gain=%4.1f bias=%4.1f  max err %.2e  %s"
          % (it, D, H, G, per, Lq, M, use_rows, gain, bias, float(err.max()), "ok" if ok else "FAIL"))
    if not ok:
        sys.exit(1)
print("all %d cases ok, worst abs err %.2e" % (cases, worst))
