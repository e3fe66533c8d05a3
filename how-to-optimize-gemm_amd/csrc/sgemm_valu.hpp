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

// Two tile configurations of the one template: <128,128,32> (8x8 outputs per thread: the rung
// BASELINE config 2 names) and <64,64,64> (4x4 per thread, 64-deep K-slices) for shapes with fewer
// 128x128 tiles than CUs -- N = 1024 has 64 of them for 256 CUs; the small tile fills the chip at
// the price of twice the LDS reads per FMA.  Same chain per element, same bits.
//
// NBUF = 2: two LDS slices, one barrier per slice (64 KiB for either tile: two workgroups per CU = two waves
// per SIMD).  NBUF = 1: one slice, the next one parked in registers across "barrier, store, barrier" (32 KiB:
// the register file becomes the limit -- three waves per SIMD at 160 registers, five for the 64x64 tile); a
// workgroup's barrier gap is filled by the CU's other workgroups.
template <int BM, int BN, int KB, bool EDGE, int NBUF>
__global__ void __launch_bounds__(256)
sgemm_valu_kernel(int m, int n, int k, const float *__restrict__ A, int lda,
                  const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                  int accumulate, int nbm, int nbn) {
  constexpr int THREADS = 256;
  constexpr int RI = BM / 64, RJ = BN / 64;        // 4x4 output blocks per thread, 64 apart
  constexpr int TI = 4 * RI, TJ = 4 * RJ;
  constexpr int A_FLOATS = KB * BM, B_FLOATS = KB * BN;
  static_assert((BM == 64 || BM == 128) && (BN == 64 || BN == 128), "thread map: 16 x 16 threads of 4x4 blocks");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  int tm, tn;
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;

  float acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int row = row0 + 4 * ty + (i & 3) + 64 * (i >> 2);
      const int col = col0 + 4 * tx + (j & 3) + 64 * (j >> 2);
      acc[i][j] = (accumulate && (!EDGE || (row < m && col < n)))
                      ? C[(size_t)row * ldc + col] : 0.0f;
    }

  // Make the accumulators' initial values ARRIVE before the K loop starts.  Without this hipcc's wait-count pass
  // carries "these registers may still be in flight" into the rolled k loop and puts `s_waitcnt vmcnt(0)` in
  // front of the first FMA of EVERY slice -- where it also waits for the next slice's global loads, issued a few
  // instructions earlier, to come back from L2 (seen in the round-2 build's ISA, tools/valu_isa.sh; N = 2048: 68.8 -> 76.4
  // TFLOP/s, N = 4096 with two workgroups per CU to hide it: 88.1 -> 89.4).
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; j += 4) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1]), "+v"(acc[i][j + 2]), "+v"(acc[i][j + 3]));

  Stage<BM, BN, THREADS, false, false, KB> st;
  const int nk = (k + KB - 1) / KB;
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
      if (EDGE) st.load_edge(A, lda, B, ldb, row0, col0, (kt + 1) * KB, m, n, k, tid);
      else      st.load(A, lda, B, ldb, row0, col0, (kt + 1) * KB, tid);
    }
    const float *As = lds + (NBUF == 2 ? cur : 0) * (A_FLOATS + B_FLOATS);
    const float *Bs = As + A_FLOATS;
#pragma unroll 4
    for (int kk = 0; kk < KB; ++kk) {
      const int g = swz_slot(kk >> 2);
      float a[TI], b[TJ];
#pragma unroll
      for (int h = 0; h < RI; ++h) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(As + kk * BM + 4 * ((ty + 16 * h) ^ g));
        a[4 * h] = v[0]; a[4 * h + 1] = v[1]; a[4 * h + 2] = v[2]; a[4 * h + 3] = v[3];
      }
#pragma unroll
      for (int h = 0; h < RJ; ++h) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(Bs + kk * BN + 64 * h + 4 * tx);
        b[4 * h] = v[0]; b[4 * h + 1] = v[1]; b[4 * h + 2] = v[2]; b[4 * h + 3] = v[3];
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_fmaf(a[i], b[j], acc[i][j]);
    }
    if (NBUF == 1 && more) __syncthreads();   // everybody is done reading the one slice
    if (more) {
      float *nxt = lds + (NBUF == 2 ? (cur ^ 1) : 0) * (A_FLOATS + B_FLOATS);
      st.store(nxt, nxt + A_FLOATS, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = row0 + 4 * ty + (i & 3) + 64 * (i >> 2);
#pragma unroll
    for (int jh = 0; jh < RJ; ++jh) {
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
