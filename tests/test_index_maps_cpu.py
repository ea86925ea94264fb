"""CPU restatement of the register / lane index maps behind the LDS-free epilogue of the Gram kernels
(fresco_amd/csrc/opt_fast.hip, gram_sign_epilogue): the accumulator layout of v_mfma_f32_32x32x16_f16, the operand
layouts, the constant 0 / 1 operand of the transposing products, and the v_permlane32_swap assembly of the 16-byte
pieces.  Pure numpy: what the kernel's comments claim, checked for every lane and register."""
import numpy as np


def c_layout_row(r, hi):
    """row of accumulator register r in half-wave hi (columns run along the lanes: col = lane & 31)"""
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def mfma_32x32x16(A, B):
    """A[lane][j], B[lane][j], j < 8: lane (l31, hi) supplies k = hi * 8 + j of row / column l31.  Returns D in the
    accumulator layout D[lane][r]."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for lane in range(64):
        l31, hi = lane & 31, lane >> 5
        for j in range(8):
            Am[l31, hi * 8 + j] = A[lane][j]
            Bm[hi * 8 + j, l31] = B[lane][j]
    Dm = Am @ Bm
    D = np.zeros((64, 16))
    for lane in range(64):
        l31, hi = lane & 31, lane >> 5
        for r in range(16):
            D[lane][r] = Dm[c_layout_row(r, hi), l31]
    return D


def permlane32_swap(a, b):
    """v_permlane32_swap: returns (new a, new b) per lane: lanes 0-31 get (own a, upper partner's a), lanes 32-63 (lower
    partner's b, own b)"""
    na, nb = a.copy(), b.copy()
    na[32:] = b[:32]
    nb[:32] = a[32:]
    return na, nb


def piece_of(w):
    """w[r4][lane] = 4 consecutive entries at offset 8 r4 + 4 hi of a line of 32 -> 16 contiguous entries per lane"""
    P0, P1 = permlane32_swap(w[0], w[2])
    Q0, Q1 = permlane32_swap(w[1], w[3])
    return [P0, P1, Q0, Q1]


def test_gram_epilogue_register_maps():
    rng = np.random.default_rng(0)
    S = rng.integers(-1, 2, size=(32, 32)).astype(np.float64)  # S[p][q] of one 32 x 32 block
    # accumulator layout: lane (q, hi) holds rows p = c_layout_row(r, hi)
    acc = np.array([[S[c_layout_row(r, lane >> 5), lane & 31] for r in range(16)] for lane in range(64)])
    # --- mirror position: dword r4 of a lane = rows 8 r4 + 4 hi .. + 3 of column q; after the swaps lanes 0-31 hold rows
    # 0-15 and lanes 32-63 rows 16-31 of their column
    w = [np.array([acc[lane][4 * r4: 4 * r4 + 4] for lane in range(64)]) for r4 in range(4)]
    pc = piece_of(w)
    for lane in range(64):
        q, hi = lane & 31, lane >> 5
        got = np.concatenate([pc[d][lane] for d in range(4)])
        assert np.array_equal(got, S[hi * 16: hi * 16 + 16, q]), lane
    # --- direct position: transposition on the matrix pipe.  A operand of step t = the lane's registers 8 t .. 8 t + 7;
    # B operand: lane (n, hi'), slot j -> 1 where 16 t + 8 (j >> 2) + 4 hi' + (j & 3) == n
    D = np.zeros((64, 16))
    for t in range(2):
        A = acc[:, 8 * t: 8 * t + 8]
        B = np.array([[1.0 if 16 * t + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3) == (lane & 31) else 0.0 for j in range(8)]
                      for lane in range(64)])
        D += mfma_32x32x16(A, B)
    for lane in range(64):
        p, hi = lane & 31, lane >> 5
        for r in range(16):
            assert D[lane][r] == S[p, c_layout_row(r, hi)], (lane, r)  # lanes along p, registers along q
    wd = [np.array([D[lane][4 * r4: 4 * r4 + 4] for lane in range(64)]) for r4 in range(4)]
    pd = piece_of(wd)
    for lane in range(64):
        p, hi = lane & 31, lane >> 5
        got = np.concatenate([pd[d][lane] for d in range(4)])
        assert np.array_equal(got, S[p, hi * 16: hi * 16 + 16]), lane


def test_sv_dot_operand_offsets():
    """<V, dV> epilogue of sv16b_kernel: the LDS offset a lane reads for accumulator register r equals the position of
    (channel row, pixel) in the tiled channel-major copy: [128 channel rows][4 swizzled 16-byte units of 8 pixels]"""
    for hf_wm_mi in range(4):
        rl0 = hf_wm_mi * 32
        for lane in range(64):
            l31, hi = lane & 31, lane >> 5
            lb0 = hi * 256 + ((((l31 >> 3) ^ hi) & 3) << 4) + (l31 & 7) * 2
            lb1 = lb0 ^ 32
            for r in range(16):
                off = (lb1 if ((r >> 2) & 1) else lb0) + ((r & 3) + 8 * (r >> 2)) * 64
                lr = (r & 3) + 8 * (r >> 2) + 4 * hi
                rl = rl0 + lr
                want = lr * 64 + ((((l31 >> 3) ^ (rl >> 2)) & 3) << 4) + (l31 & 7) * 2
                assert off == want, (rl0, lane, r)


def _gx_tile(idx, hw):
    """tile walk of gram16y_kernel (gx_tile in opt_fast.hip): tiles (ti, tj) of 256 x 128 pixels, tj >= 2 ti in 128-pixel units"""
    if hw % 1024 == 0:
        ns = hw // 1024
        si = 0
        while True:
            row_tiles = 20 + 32 * (ns - 1 - si)
            if idx < row_tiles:
                break
            idx -= row_tiles
            si += 1
        if idx < 20:
            r = 0
            while idx >= 8 - 2 * r:
                idx -= 8 - 2 * r
                r += 1
            return si * 4 + r, si * 8 + 2 * r + idx
        idx -= 20
        sj = si + 1 + idx // 32
        return si * 4 + (idx % 32) // 8, sj * 8 + idx % 8
    n128 = hw // 128
    ti = 0
    while idx >= n128 - 2 * ti:
        idx -= n128 - 2 * ti
        ti += 1
    return ti, 2 * ti + idx


def test_gram_tile_walks_write_every_block_once():
    """Both Gram kernels must write every 64 x 64 block of the sign matrix exactly once -- directly or as the mirror
    image of the block across the diagonal (blocks ON the diagonal of a 128-pixel unit are computed whole)."""
    for hw in (512, 768, 1024, 1536, 2048, 2304, 4096):
        n64, n128, n256 = hw // 64, hw // 128, hw // 256
        # --- gram16y: 256 x 128 tiles, 8 waves as 4 x 2
        cnt = np.zeros((n64, n64), dtype=int)
        ntiles = n256 * n128 - n256 * (n256 - 1)
        seen = set()
        for idx in range(ntiles):
            ti, tj = _gx_tile(idx, hw)
            assert 0 <= ti < n256 and 2 * ti <= tj < n128 and (ti, tj) not in seen, (hw, idx, ti, tj)
            seen.add((ti, tj))
            for wm in range(4):
                a_sub = 2 * ti + (wm >> 1)
                wgt = 0 if a_sub > tj else (2 if a_sub < tj else 1)
                for wn in range(2):
                    r, c = ti * 4 + wm, tj * 2 + wn  # 64-pixel block coordinates
                    if wgt >= 1:
                        cnt[r, c] += 1
                    if wgt == 2:
                        cnt[c, r] += 1
        assert (cnt == 1).all(), (hw, "gram16y", int((cnt != 1).sum()))
        # --- gram16z: 128 x 128 tiles (ti, tj >= ti), 4 waves as 2 x 2
        cnt = np.zeros((n64, n64), dtype=int)
        idx = 0
        for ti in range(n128):
            for tj in range(ti, n128):
                # the kernel's walk: idx -> (ti, tj) row by row
                t, rem = 0, idx
                while rem >= n128 - t:
                    rem -= n128 - t
                    t += 1
                assert (t, t + rem) == (ti, tj)
                idx += 1
                for wm in range(2):
                    for wn in range(2):
                        r, c = ti * 2 + wm, tj * 2 + wn
                        cnt[r, c] += 1
                        if ti < tj:
                            cnt[c, r] += 1
        assert idx == n128 * (n128 + 1) // 2
        assert (cnt == 1).all(), (hw, "gram16z", int((cnt != 1).sum()))
