// sgemm_mfma.hpp -- K2: the MI355X SGEMM hot kernel.
//
// What it computes is what cuda/MMult_cuda_12.cu:86-223 (sgemm_128x128x8)
// computes for the reference -- one C tile per workgroup, fp32, row-major,
// each C(i,j) a single fp32 accumulator fed k products in ascending k -- but
// the machinery is CDNA4's: the accumulators are MFMA tiles
// (v_mfma_f32_16x16x4_f32: D = A[16x4]*B[4x16] + C, bit-for-bit an fmaf chain
// over its 4 k's in order), A/B K-slices are packed into LDS by
// sgemm_tile.hpp, and one ds_read_b128 per operand feeds 16 MFMAs.
//
// Geometry: BM x BN block tile (128x128: 4 waves as 2x2; 256x128: 8 waves as
// 4x2), every wave owns a 64x64 sub-tile = 4x4 MFMA tiles = 64 accumulator
// VGPRs.  K-slices of BK=32 are double-buffered in LDS: while the MFMAs chew
// on buffer `cur`, the next slice's global loads are in flight into
// registers, then written to buffer `cur^1`; one barrier per K-slice
// (functional twin of the ldg/sts ping-pong at cuda/MMult_cuda_12.cu:151-208).
//
// Per K-slice and wave: 8 k-steps x 16 MFMA = 128 MFMAs = 4096 matrix-pipe
// cycles against 16 ds_read_b128, 8 global_load_dwordx4 and 8 ds_write_b128.
//
// Row interleave: MFMA tile t of a wave covers rows {m0 + 4i + t}; the D
// layout (lane l, reg r -> tile row 4*(l>>4)+r, tile col l&15) then puts four
// CONSECUTIVE columns n0+4*(l&15)+{0..3} of one C row in the same lane across
// the four column tiles, so the epilogue is global_store_dwordx4.
#pragma once
#include "sgemm_tile.hpp"

namespace mmh {

template <int BM, int BN, bool EDGE>
__global__ void __launch_bounds__(BM * BN / (64 * 64) * 64)
sgemm_mfma_kernel(int m, int n, int k, const float *__restrict__ A, int lda,
                  const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                  int accumulate, int nbm, int nbn) {
  constexpr int WAVES_N = BN / 64;
  constexpr int THREADS = BM * BN / (64 * 64) * 64;
  constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * BN;
  // one LDS object (dynamic, sized by the launcher): [buf][A slice | B slice]
  extern __shared__ __attribute__((aligned(16))) float lds[];

  int tm, tn;
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15;   // MFMA row/col index within a tile
  const int kq = lane >> 4;   // which of the 4 k's of a k-step this lane feeds

  // C rows/cols this lane owns: row(t, r) = crow + 4r + t, cols ccol..ccol+3
  const int crow = row0 + wm * 64 + 16 * kq;
  const int ccol = col0 + wn * 64 + 4 * li;

  f32x4 acc[4][4];
  if (accumulate) {
    // C's current value is the first term of each element's chain, as in
    // armv7/REF_MMult.c:18 (C(i,j) = C(i,j) + A(i,p)*B(p,j)).
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = crow + 4 * r + t;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!EDGE) {
          v = *reinterpret_cast<const f32x4 *>(C + (size_t)row * ldc + ccol);
        } else if (row < m) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (ccol + u < n) v[u] = C[(size_t)row * ldc + ccol + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u][r] = v[u];
      }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  Stage<BM, BN, THREADS> st;
  const int nk = (k + BK - 1) / BK;

  // fragment read offsets (floats) inside a buffer, without the per-k-step part
  const int a_slot = wm * 16 + li;            // slot = m/4 before swizzle
  const int b_off = A_FLOATS + kq * BN + wn * 64 + 4 * li;

  if (nk > 0) {
    if (EDGE) st.load_edge(A, lda, B, ldb, row0, col0, 0, m, n, k, tid);
    else      st.load(A, lda, B, ldb, row0, col0, 0, tid);
    st.store(lds, lds + A_FLOATS, tid);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      if (EDGE) st.load_edge(A, lda, B, ldb, row0, col0, (kt + 1) * BK, m, n, k, tid);
      else      st.load(A, lda, B, ldb, row0, col0, (kt + 1) * BK, tid);
    }
    const float *buf = lds + cur * (A_FLOATS + B_FLOATS);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(
          buf + (4 * ks + kq) * BM + 4 * (a_slot ^ swz_slot(ks)));
      const f32x4 b = *reinterpret_cast<const f32x4 *>(buf + b_off + 4 * ks * BN);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
    if (more) {
      float *nxt = lds + (cur ^ 1) * (A_FLOATS + B_FLOATS);
      st.store(nxt, nxt + A_FLOATS, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: 16 x global_store_dwordx4 per lane (256 B contiguous per 16 lanes)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 4 * r + t;
      f32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
      if (!EDGE) {
        *reinterpret_cast<f32x4 *>(C + (size_t)row * ldc + ccol) = v;
      } else if (row < m) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ccol + u < n) C[(size_t)row * ldc + ccol + u] = v[u];
      }
    }
}

}  // namespace mmh
