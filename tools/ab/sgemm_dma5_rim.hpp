// sgemm_dma5_rim.hpp -- round 4's FUSED rim of the K2W 64x64 tile (tools build only; moved out of csrc/sgemm_dma5.hpp in
// round 5: the product's kernel headers hold what ships).  Included by csrc/launch_dma5.hip under -DMMH_AB_BUILD.
#pragma once
#include "sgemm_dma5.hpp"

namespace mmh {

template <class S>
struct Dma5RimWave {
  using T = typename S::T;
  static constexpr int RIM_BASE = S::RIM_BASE, RIM_STRIDE = S::RIM_STRIDE;
  static constexpr int BM = 64, BN = 64, KB = 32, NBUF = T::LA + 1;   // (Dma5Segment: the rim rides on the guarded 64x64 tile)
  // ---------------------------------------------------------------------------------------------------- the rim wave
  // A shape one or two elements past a 64-boundary (N = 1025) would pay a whole extra row and column of tiles -- thin
  // ones since round 4, but their MFMAs (a 16-row block for one valid row) still land on 33 of 256 CUs and set the
  // launch's time.  A RIM launch runs the tiles of the TRIMMED shape, and every tile of the last tile row / column
  // computes ITS 64 elements of each rim row / column in one extra wave, on the vector ALU, out of the K-slices its
  // workgroup has in LDS anyway: lane j <-> column col0 + j of a rim row (B's slice image: a conflict-free ds_read_b32
  // per k), lane i <-> row row0 + i of a rim column (A's row-major image: a ds_read_b128 per four k); the other operand
  // -- the rim row's A values, the rim column's B values, 128 and 512 bytes per slice -- comes in by two extra LDS-DMA
  // pieces (the loaders, above) and is read as a broadcast.  One v_fma_f32 per element and k, in ascending k: the
  // chain the MFMA computes, bit for bit (what makes K1 and K2 agree).  The corner tile's lanes 0 .. r_m - 1 do the
  // corner elements the same way.  The wave meets the workgroup at every slice barrier (it reads the ring like a
  // consumer: everything of slice kt between the barriers of slices kt - 1 and kt).
  // MEASURED (profiles/r04_notes.md; tools build only, MMH_OPT_RIM5): correct to the bit -- and 2.2x SLOWER per edge tile
  // than a whole tile (N = 1025: K loop 43.6 us against 19.9 us for the interior tiles of the same launch, 46 TFLOP/s
  // against 83 for the thin edge tiles).  The f32 MFMA runs on the vector ALU's own FMA lanes ("at the f32 VECTOR rate"):
  // on a SIMD whose matrix pipe a consumer keeps busy, the rim wave's v_fma_f32 get one issue slot per MFMA -- 96
  // dependent VALU operations per slice at ~28 cycles each.  A rim on the vector ALU cannot hide beside an MFMA loop;
  // the thin edge tiles (their MFMAs cost 16 rows for one) stay the product's answer for N + 1.
  static __device__ __forceinline__ void run(float *lds, int m, int n, int k, float *__restrict__ C, int ldc, int row0, int col0,
                                                  int kb, int ke, bool init_from_c, const Dma5Rim rim) {
    // three copies of the loop, picked by what this tile has (wave-uniform): the rim row only, the rim column only, both
    // (the corner tile) -- each reads and multiplies only what it needs
    if (rim.do_row && rim.do_col) rim_pass(std::true_type{}, std::true_type{}, lds, k, C, ldc, row0, col0, kb, ke, init_from_c, rim);
    else if (rim.do_row) rim_pass(std::true_type{}, std::false_type{}, lds, k, C, ldc, row0, col0, kb, ke, init_from_c, rim);
    else rim_pass(std::false_type{}, std::true_type{}, lds, k, C, ldc, row0, col0, kb, ke, init_from_c, rim);
  }

  template <class ROW_T, class COL_T>
  static __device__ __forceinline__ void rim_pass(ROW_T, COL_T, float *lds, int k, float *__restrict__ C, int ldc, int row0, int col0,
                                                  int kb, int ke, bool init_from_c, const Dma5Rim rim) {
    constexpr bool ROW = ROW_T::value, COL = COL_T::value, CORNER = ROW && COL;
    constexpr int STAGE = T::STAGE, A_FLOATS = T::A_FLOATS, NC = KB / 4;
    const int lane = threadIdx.x & 63;
    // rim row: (m0, col0 + lane); rim column: (row0 + lane, n0); corner: (m0, n0), every lane alike, lane 0 stores
    float acc_r = 0.f, acc_c = 0.f, acc_x = 0.f;
    const int col = col0 + lane, row = row0 + lane;
    const bool col_ok = col < rim.n0, row_ok = row < rim.m0;
    if (init_from_c) {
      if (ROW && col_ok) acc_r = C[(size_t)rim.m0 * ldc + col];
      if (COL && row_ok) acc_c = C[(size_t)row * ldc + rim.n0];
      if (CORNER) acc_x = C[(size_t)rim.m0 * ldc + rim.n0];
    }
    const int a_row = lane * KB, a_swz = lane & 7;              // this lane's row of A's image, its chunk XOR
    const int b_col = 4 * (lane >> 2), b_in = lane & 3;         // this lane's column of B's image
    __builtin_amdgcn_s_barrier();   // the loaders have the first slice in LDS
    int pos = 0;
    for (int kt = kb; kt < ke; ++kt) {
      const float *buf = lds + pos * STAGE;
      const float *ra = lds + RIM_BASE + pos * RIM_STRIDE, *rb = ra + 256;
      const int kvalid = __builtin_amdgcn_readfirstlane(min(KB, k - kt * KB));   // (the problem's last slice may be partial)
      // every operand of the slice first (one LDS round trip per slice, not per k), then the chains
      f32x4 a_r[NC], a_own[NC], b_c[NC];   // rim row's A values (broadcast), this lane's A row, rim column's B values (broadcast)
      float b_own[KB];                     // this lane's B column
      static_for<NC>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        if constexpr (ROW) a_r[c] = *reinterpret_cast<const f32x4 *>(ra + 4 * c);
        if constexpr (COL) {
          a_own[c] = *reinterpret_cast<const f32x4 *>(buf + a_row + 4 * (c ^ a_swz));
          b_c[c] = *reinterpret_cast<const f32x4 *>(rb + 4 * c);
        }
      });
      if constexpr (ROW) {
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) b_own[kk] = buf[A_FLOATS + kk * BN + (b_col ^ ((kk & 1) << 5)) + b_in];   // (chunk XOR 8 = float offset XOR 32)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // slice kt's barrier: every read of this ring position has completed
      // (k's past the end -- the problem's last, partial slice -- have their OPERANDS zeroed, as the consumers' tail slice
      // does: A's columns there hold the next row or the caller's padding; the selects stay off the accumulators' chains)
      static_for<NC>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool live = 4 * c + q < kvalid;   // wave-uniform
          if constexpr (ROW) acc_r = __builtin_fmaf(live ? a_r[c][q] : 0.0f, live ? b_own[4 * c + q] : 0.0f, acc_r);
          if constexpr (COL) acc_c = __builtin_fmaf(live ? a_own[c][q] : 0.0f, live ? b_c[c][q] : 0.0f, acc_c);
          if constexpr (CORNER) acc_x = __builtin_fmaf(live ? a_r[c][q] : 0.0f, live ? b_c[c][q] : 0.0f, acc_x);
        }
      });
      pos = pos == NBUF - 1 ? 0 : pos + 1;
    }
    if (ROW && col_ok) C[(size_t)rim.m0 * ldc + col] = acc_r;
    if (COL && row_ok) C[(size_t)row * ldc + rim.n0] = acc_c;
    if (CORNER && lane == 0) C[(size_t)rim.m0 * ldc + rim.n0] = acc_x;
  }

};

// The RIM launch (Dma5Segment::rim_wave): one workgroup per tile of the TRIMMED shape (m - r_m) x (n - r_n), r_m, r_n <= 1
// row / column past a 64-boundary; four consumers, two loaders and the rim wave.  Interior tiles' rim waves leave at
// once.  Whole tiles all: the raster is the trimmed grid's.
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, int NL, int D>
__global__ void __launch_bounds__(64 * (5 + NL))
sgemm_mfma_dma5_rim_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                           float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn, int r_m, int r_n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = Dma5Segment<BM, BN, KB, WTM, WTN, NBUF, false, true, false, NL, D, true>;
  int tm, tn;
  dma_stamp(0);
  block_to_tile_g(blockIdx.x, nbm * nbn, nbm, nbn, Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>::GM, tm, tn);
  Dma5Rim rim;
  rim.m0 = m - r_m;
  rim.n0 = n - r_n;
  rim.r_m = r_m;
  rim.r_n = r_n;
  rim.do_row = r_m > 0 && tm == nbm - 1;
  rim.do_col = r_n > 0 && tn == nbn - 1;
  typename S::Lane L;
  L.init(lda, ldb);
  if (L.rim && !rim.do_row && !rim.do_col) return;   // (before any barrier: a wave that has ended is not waited for)
  typename S::Frags fr;
  Dma5Link link;
  int no_reply = 0;
  S::run(lds, L, rim.m0, rim.n0, k, A, lda, B, ldb, C, ldc, tm, tn, 0, (k + KB - 1) / KB, accumulate != 0, nullptr, nullptr, fr,
         link, Dma5Next{}, nullptr, no_reply, rim);
  dma_stamp_after_stores(3);
}

}  // namespace mmh
