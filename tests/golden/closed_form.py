"""Closed-form (RNG-free) test inputs shared by the fixture generator and the parity tests.

Formulas are those of SURVEY.md Appendix B; every tensor is built from float64 index grids and
cast to float32, so the build container and the GPU box construct bit-identical inputs.
"""
import math

import torch


def _grid(B, C, h, w):
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    c = torch.arange(C, dtype=torch.float64).view(1, C, 1, 1)
    i = torch.arange(h, dtype=torch.float64).view(1, 1, h, 1)
    j = torch.arange(w, dtype=torch.float64).view(1, 1, 1, w)
    return b, c, i, j


def feat(B, C, h, w, ph):
    b, c, i, j = _grid(B, C, h, w)
    return torch.sin(0.1 * (b + 1) * (c + 1) + 0.37 * i + 0.11 * j + ph).float()


def flow(B, H, W, s):
    b, _, i, j = _grid(B, 1, H, W)
    fx = s * (1.5 * torch.sin(0.02 * i) + 0.5 * b + 1) + 0 * j
    fy = s * (-torch.cos(0.03 * j) - 0.25 * b) + 0 * i
    return torch.cat([fx, fy], 1).float()


def occ(B, H, W, k):
    b, _, i, j = _grid(B, 1, H, W)
    v = (torch.floor(i / 8) + torch.floor(j / 8) + b) % k == 0
    return v[:, 0].float()


def base_case(N=4, H=64, W=64):
    """The N=4, 64x64 case of Appendix B."""
    d = {}
    d["fwd"] = flow(N, H, W, +1.0)
    d["bwd"] = flow(N, H, W, -1.0)
    d["x"] = feat(N, 3, H, W, 0.0)
    d["imgs"] = feat(N, 3, H, W, 0.5)
    d["fo"] = occ(N, H, W, 5)
    d["bo"] = occ(N, H, W, 7)
    d["sal"] = torch.sigmoid(feat(N, 1, H // 2, W // 2, 1.0))
    d["lat"] = feat(2 * N, 8, 8, 8, 0.2)
    return d


def attn_weights(C):
    """W_k[i,j] = sin(0.05 i (k+1) + 0.07 j + k)/sqrt(C), k = 0..3 for (q,k,v,out); out bias 0."""
    i = torch.arange(C, dtype=torch.float64).view(C, 1)
    j = torch.arange(C, dtype=torch.float64).view(1, C)
    return [(torch.sin(0.05 * i * (k + 1) + 0.07 * j + k) / math.sqrt(C)).float() for k in range(4)]


def attn_hidden(B, HW, C, phase=0.0):
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1)
    p = torch.arange(HW, dtype=torch.float64).view(1, HW, 1)
    c = torch.arange(C, dtype=torch.float64).view(1, 1, C)
    return torch.sin(0.3 * (b + 1) + 0.05 * p + 0.21 * c + phase).float()


def checksum(t):
    t = t.double()
    return float(t.sum()), float(t.abs().sum())


def video_case(N, H, W):
    """uint8 HxWx3 frames (as cv2 hands them to get_flow_and_interframe_paras) + closed-form flows for
    the pairs (i -> i+1 mod N).  Returns (list of numpy frames, fwd flows, bwd flows)."""
    img = feat(N, 3, H, W, 0.5).double()
    u8 = torch.round((img * 0.5 + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
    frames = [u8[i].permute(1, 2, 0).contiguous().numpy() for i in range(N)]
    return frames, flow(N, H, W, +1.0), flow(N, H, W, -1.0)


def gmflow_param(name, shape):
    """Closed-form stand-in weights for the flow network (the published checkpoint is not available):
    a deterministic function of the parameter's NAME and SHAPE only, so that the reference model (fixture
    generation) and fresco_amd.gmflow (GPU test) are filled identically without shipping a state dict.
    Matrices / kernels ~ unit-variance / sqrt(fan_in); LayerNorm scales ~ 1; biases small."""
    n = 1
    for s in shape:
        n *= int(s)
    seed = sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 9973
    i = torch.arange(n, dtype=torch.float64)
    base = torch.sin(i * 0.7548776662466927 + 0.001 * seed) * torch.cos(i * 0.5698402909980532 + 0.013 * seed)
    if len(shape) > 1:
        fan_in = n // int(shape[0])
        val = base * (2.0 / math.sqrt(fan_in))
    elif name.endswith("norm1.weight") or name.endswith("norm2.weight"):
        val = 1.0 + 0.1 * base
    else:
        val = 0.05 * base
    return val.reshape(shape).float()


def gmflow_frames(N, H, W):
    """N float frames (N,3,H,W) in 0..255: one smooth texture translated by (2, -3) px per frame"""
    out = []
    for n in range(N):
        _, c, i, j = _grid(1, 3, H, W)
        ii, jj = i + 2.0 * n, j - 3.0 * n
        v = (torch.sin(0.21 * ii + 0.4 * c) * torch.cos(0.17 * jj - 0.3 * c) + 0.5 * torch.sin(0.05 * ii * (c + 1) + 0.09 * jj)
             + 0.3 * torch.sin(0.9 * ii) * torch.sin(0.8 * jj))
        out.append((v * 70.0 + 128.0).clamp(0, 255)[0])
    return torch.stack(out, 0).float()
