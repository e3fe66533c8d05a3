// sgemm_valu.hpp -- K1: LDS-tiled SGEMM on the vector ALU only (no MFMA), and
// K0: the naive one-thread-per-element kernel.
//
// K1 is BASELINE.json config 2 ("LDS-tiled no-MFMA baseline"): the MI355X
// counterpart of the reference's shared-memory ladder
// cuda/MMult_cuda_3.cu:10-53 ... cuda/MMult_cuda_9.cu:30-125 -- 128x128 block
// tile, 256 threads, 8x8 outputs per thread as a 2x2 arrangement of 4x4
// blocks 64 apart (cf. cuda/MMult_cuda_9.cu:71-96), outer-product inner loop.
// It shares the packing stage with the MFMA kernel (sgemm_tile.hpp), so the
// A/B fragments are ds_read_b128 and every thread of a 16-lane group reads
// distinct 16-byte slots of B while A is a broadcast.
// Each C(i,j) is one fmaf chain over ascending k: bit-identical to K2.
//
// K0 mirrors cuda/MMult_cuda_2.cu:12-35 (the only reference kernel with a
// bounds check); it exists as the bottom rung of the ladder and as an
// independent on-device cross-check of K1/K2.
#pragma once
#include "sgemm_tile.hpp"

namespace mmh {

template <bool EDGE>
__global__ void __launch_bounds__(256)
sgemm_valu_kernel(int m, int n, int k, const float *__restrict__ A, int lda,
                  const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                  int accumulate, int nbm, int nbn) {
  constexpr int BM = 128, BN = 128, THREADS = 256;
  constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * BN;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  int tm, tn;
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;

  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = row0 + 4 * ty + (i & 3) + 64 * (i >> 2);
      const int col = col0 + 4 * tx + (j & 3) + 64 * (j >> 2);
      acc[i][j] = (accumulate && (!EDGE || (row < m && col < n)))
                      ? C[(size_t)row * ldc + col] : 0.0f;
    }

  Stage<BM, BN, THREADS> st;
  const int nk = (k + BK - 1) / BK;
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
    const float *As = lds + cur * (A_FLOATS + B_FLOATS);
    const float *Bs = As + A_FLOATS;
#pragma unroll 4
    for (int kk = 0; kk < BK; ++kk) {
      const int g = swz_slot(kk >> 2);
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(As + kk * BM + 4 * (ty ^ g));
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(As + kk * BM + 4 * ((ty + 16) ^ g));
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bs + kk * BN + 4 * tx);
      const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bs + kk * BN + 64 + 4 * tx);
      const float a[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const float b[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_fmaf(a[i], b[j], acc[i][j]);
    }
    if (more) {
      float *nxt = lds + (cur ^ 1) * (A_FLOATS + B_FLOATS);
      st.store(nxt, nxt + A_FLOATS, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = row0 + 4 * ty + (i & 3) + 64 * (i >> 2);
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      const int col = col0 + 4 * tx + 64 * jh;
      if (!EDGE) {
        f32x4 v = {acc[i][4 * jh], acc[i][4 * jh + 1], acc[i][4 * jh + 2], acc[i][4 * jh + 3]};
        *reinterpret_cast<f32x4 *>(C + (size_t)row * ldc + col) = v;
      } else if (row < m) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (col + u < n) C[(size_t)row * ldc + col + u] = acc[i][4 * jh + u];
      }
    }
  }
}

// K0: one thread per C element, row-major, coalesced along n.
__global__ void __launch_bounds__(256)
sgemm_naive_kernel(int m, int n, int k, const float *__restrict__ A, int lda,
                   const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                   int accumulate) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= m || col >= n) return;
  float acc = accumulate ? C[(size_t)row * ldc + col] : 0.0f;
  for (int p = 0; p < k; ++p)
    acc = __builtin_fmaf(A[(size_t)row * lda + p], B[(size_t)p * ldb + col], acc);
  C[(size_t)row * ldc + col] = acc;
}

}  // namespace mmh
