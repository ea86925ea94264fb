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
