// moments_mid_dev.hpp -- the half-tile layout shared by the streaming kernels of 17 .. 64 features: moments_mid.hip (one regression,
// fused report), grouped_mid.hip (groups: records, or the paired stream with its solving waves), leverage_mid.hip follows the same scheme.
#pragma once
#include "common.hpp"

namespace pds {
namespace {


constexpr int kMidWavesPerCu = 4;

// NBLK = 2 (17 .. 32 features) or 4 (33 .. 64).  One asynchronous load instruction moves 1 KiB: NBLK columns x HR rows -- the 64 / NBLK
// lanes of lane group g fetch 16 bytes each of column 16 g + i -- so a half-tile is 16 such instructions + one for the target.
// (One column per instruction with half or a quarter of the lanes active: the instruction count, not the bytes, was the limit --
// ~90 clk of the wave's time per global_load_lds whatever it moves.)  The LDS image keeps an instruction's 1 KiB together: column
// (b, i) of the half-tile sits at i * GS + b * HR * 8, GS = 1024 + 16.  The operand fetch of block b reads 16 consecutive i at one b:
// bank = (4 i + 2 row) mod 64 -- distinct within a half-wave.
template <int NBLK, int ES = 8 /* element bytes: f64 frames; 4 = f32 frames, widened to f64 on their way out of LDS */>
struct MidDims {
    static constexpr int HR = 1024 / (NBLK * ES);              // rows per half-tile (128 / NBLK at f64, 256 / NBLK at f32)
    static constexpr int GL = 64 / NBLK;                       // lanes per lane group = 16-byte pieces per column
    static constexpr int GS = 1024 + 16;                       // bytes between the images of instructions i and i + 1
    static constexpr int Y_OFF = 16 * GS;                      // the target column's image (HR * ES bytes)
    static constexpr int W_OFF = 16 * GS + HR * ES + 16;       // the weight column's image (weighted form)
    static constexpr int HALF_BYTES = 16 * GS + 2 * (HR * ES + 16) + 16;
    static constexpr int NBUF = 2;
    static constexpr int LDS_BYTES = NBUF * HALF_BYTES;
    static constexpr int NS = HR / 4;                          // 4-row steps per half-tile
    static constexpr int NPAIR = NBLK * (NBLK + 1) / 2;
    // per-wave partial record (doubles): NPAIR tiles of 4 registers x 64 lanes | xy, cs: NBLK x 64 each | yy, ys, sw: 64 each
    static constexpr int REC = NPAIR * 256 + 2 * NBLK * 64 + 256;  // (+ sum w, weighted form; + sum e^2, fused report form)
};

// Fused report form (FUSE = 1: HC0 / HC1, 2: HC2 / HC3): four waves per workgroup share a front of the LDS --
//   [ FUSE == 2: operand blocks of L | row p of L | L[p][p]^2, 0 ]  (leverage_operand, leverage_mid.hip)   | beta (16 NBLK) | b0, 0
template <int NBLK, int FUSE>
struct MidShared {
    static constexpr int NLB = 4 * (NBLK * (NBLK + 1) / 2);
    static constexpr int LB_OFF = FUSE == 2 ? NLB * 512 : 0;
    static constexpr int C0_OFF = LB_OFF + (FUSE == 2 ? 16 * NBLK * 8 : 0);
    static constexpr int BETA_OFF = C0_OFF + (FUSE == 2 ? 16 : 0);
    static constexpr int BYTES = FUSE ? BETA_OFF + (16 * NBLK + 2) * 8 : 0;
    static constexpr int WPB = FUSE ? 4 : 1;  // waves per workgroup
};

// operand block (ablk, kstep), kstep >= 4 ablk (the layout of leverage_operand)
template <int NBLK>
__device__ constexpr int mid_lev_block(int ablk, int kstep) {
    int base = 0;
    for (int a = 0; a < ablk; ++a) base += 4 * NBLK - 4 * a;
    return base + (kstep - 4 * ablk);
}

}  // namespace
}  // namespace pds
