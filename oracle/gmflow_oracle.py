"""TEST INFRASTRUCTURE -- not part of the product.

CPU restatement of GMFlow's (shifted) split-window attention as the reference computes it
(src/ebsynth/deps/gmflow/gmflow/transformer.py:20-108, utils.py:5-51): roll by half a window, cut into
num_splits^2 windows, add the -100 mask between tokens of different pre-roll regions, softmax, merge, roll back.
fresco_amd.gmflow replaces all of this by token groups; tests/test_gmflow.py checks the two against each other.
Pinned indirectly: the end-to-end goldens of tests/golden/gmflow_golden.npz come from the unmodified reference."""
import torch


def _split(x, s):  # (b,h,w,c) -> (b*s*s, h/s, w/s, c)   utils.py:5-17
    b, h, w, c = x.shape
    return x.view(b, s, h // s, s, w // s, c).permute(0, 1, 3, 2, 4, 5).reshape(b * s * s, h // s, w // s, c)


def _merge(x, s):  # inverse of _split   utils.py:34-44
    bn, hh, ww, c = x.shape
    b = bn // (s * s)
    return x.view(b, s, s, hh, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(b, s * hh, s * ww, c)


def shift_mask(h, w, wh, ww, sh, sw):
    """transformer.py:20-44: (splits^2, wh*ww, wh*ww) additive mask, 0 inside a region, -100 across regions"""
    img = torch.zeros(1, h, w, 1)
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    mw = _split(img, w // ww).view(-1, wh * ww)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return d.masked_fill(d != 0, -100.0).masked_fill(d == 0, 0.0)


def swin_attention(q, k, v, h, w, splits, shifted):
    """q, k, v (b, h*w, c) -> (b, h*w, c); transformer.py:48-108 (full attention when splits == 1: :8-17)"""
    b, _, c = q.shape
    if splits == 1:
        s = (q @ k.transpose(1, 2)) / c ** 0.5
        return torch.softmax(s, -1) @ v
    wh, ww = h // splits, w // splits
    sh, sw = wh // 2, ww // 2
    q, k, v = (t.view(b, h, w, c) for t in (q, k, v))
    if shifted:
        q, k, v = (torch.roll(t, shifts=(-sh, -sw), dims=(1, 2)) for t in (q, k, v))
    qs, ks, vs = (_split(t, splits).reshape(b * splits * splits, -1, c) for t in (q, k, v))
    s = (qs @ ks.transpose(1, 2)) / c ** 0.5
    if shifted:
        s = s + shift_mask(h, w, wh, ww, sh, sw).to(s.dtype).repeat(b, 1, 1)
    out = torch.softmax(s, -1) @ vs
    out = _merge(out.view(b * splits * splits, wh, ww, c), splits)
    if shifted:
        out = torch.roll(out, shifts=(sh, sw), dims=(1, 2))
    return out.reshape(b, h * w, c)
