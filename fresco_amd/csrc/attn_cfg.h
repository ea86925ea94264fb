// Configuration of the dense-attention kernels of attn.hip (packed key-image geometry, softmax thresholds).  Not part of
// the public ABI.
#pragma once
#include "common.h"

namespace fresco {

template <int D>
struct AttnCfg {
    static constexpr int DPK = (D + 15) / 16 * 16;  // head dim padded for the QK^T contraction
    static constexpr int DPV = (D + 31) / 32 * 32;  // head dim padded to whole 32-row blocks of O^T
    static constexpr int NKS = DPK / 16;            // MFMA k-steps per QK^T block
    static constexpr int NDB = DPV / 32;            // 32-row blocks of O^T
    // LDS / packed image of one 64-key tile, in 16-byte chunks (8 halfs):
    //   K  : chunk ((ks*2 + c)*64 + key)   = K[key][ks*16 + c*8 .. +8]                     (c = MFMA k-chunk)
    //   V^T: chunk ((kc*2 + c)*DPV + d)    = V[kc*16 + slot(c, e)][d],  e = 0..7           (kc = 16-key MFMA step)
    //        slot(c, e) = (e & 3) + 8*(e >> 2) + 4*c: the order in which a lane's C-tile registers hold the keys
    // A 16-lane ds_read_b128 group reads 16 different keys (or 16 different d) at a chunk stride of 1 and one
    // c: conflict-free without padding.
    static constexpr int KTILE = DPK * 128;  // bytes
    static constexpr int VTILE = DPV * 128;
    static constexpr int TILE = KTILE + VTILE;
    static constexpr int NP = TILE / 1024;  // 1 KiB DMA pieces per tile
    static constexpr int NBUF = 4;          // ring slots
    static constexpr int LDS_BYTES = NBUF * TILE;
    static constexpr bool ONES = DPV > D;  // spare V^T row D holds ones: the PV MFMA also yields the row sum
    static constexpr bool MCOL = DPK > D;  // spare K column D holds ones: Q column D carries -m_run, so the
                                           // QK MFMA subtracts the running max (no C operand to keep around)
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (tools/flash_regime.py restates the two per-wave decisions these thresholds drive on the CPU)
// online softmax: skip the O rescale while the tile max grows by less than this (log2 units);
// P then reaches at most 2^8 = 256, far inside fp16 range, and stays exactly normalised by the row sum
constexpr float RESCALE_THR = 8.0f;
// no running max at all when |c q| max|k| - m_run stays below this (P <= 2^14 = 16384 < 65504)
constexpr float NOMAX_THR = 14.0f;
// largest Cauchy-Schwarz logit bound c |q| max|k| (log2 units) for which the scale is folded into the fp16 Q.  Measured
// margin (tools/fold_margin.py, CPU emulation of the kernel's arithmetic against fp64): with N(0,1) keys the folded
// form's worst error is 0.30 of the 1e-3 parity bar for bounds up to 24 (0.09 for the exact form), 0.42 up to 32; with
// keys ALIGNED to the query (logit = bound) both forms sit at the same error, set by P's own fp16 rounding.  Round 2
// used 16: N(0,1) q, k (bound up to 23) then ran the exact pass on every workgroup, 11 % slower (r03_ab_variants.txt).
constexpr float FOLD_MAX = 24.0f;

static inline int ntiles_of(int M) { return (M + 63) / 64; }

}  // namespace fresco
