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
// NBUF = 2: two LDS slices, one barrier per slice (64 KiB for either tile: two workgroups per CU).  NBUF = 1: one
// slice, the next one parked in registers across "barrier, store, barrier" (32 KiB: the register file becomes the
// limit -- two waves per SIMD for the 128x128 tile at ~190 registers, three for the 64x64 tile at 132-142); a workgroup's barrier
// gap is filled by the CU's other workgroups.
// Round 4: the K-slice is fully unrolled with the fragments of k-step kk + P requested before the FMAs of k-step kk
// (registers, not another wave, cover the LDS round trip), and the accumulators are pairs so that every FMA is a
// v_pk_fma_f32: N = 1024 31.6 -> 58 TFLOP/s (one 64x64 tile per CU = one wave per SIMD), 2048 77 -> 83, 4096 91 -> 92.
template <int BM, int BN, int KB, bool EDGE, int NBUF, int P>   // P: k-steps of fragment look-ahead
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

  // accumulators as pairs of adjacent columns: one v_pk_fma_f32 per pair and k-step (a[i] broadcast, {b[j], b[j + 1]})
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[TI][TJ / 2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int row = row0 + 4 * ty + (i & 3) + 64 * (i >> 2);
      const int col = col0 + 4 * tx + (j & 3) + 64 * (j >> 2);
      acc[i][j >> 1][j & 1] = (accumulate && (!EDGE || (row < m && col < n)))
                                  ? C[(size_t)row * ldc + col] : 0.0f;
    }

  // Make the accumulators' initial values ARRIVE before the K loop starts.  Without this hipcc's wait-count pass
  // carries "these registers may still be in flight" into the k loop and puts `s_waitcnt vmcnt(0)` in
  // front of the first FMA of EVERY slice -- where it also waits for the next slice's global loads, issued a few
  // instructions earlier, to come back from L2 (seen in the round-2 build's ISA, tools/valu_isa.sh; N = 2048: 68.8 -> 76.4
  // TFLOP/s, N = 4096 with two workgroups per CU to hide it: 88.1 -> 89.4).
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ / 2; j += 2) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1]));

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
    // The fragments of k-step kk + P are requested before the FMAs of k-step kk (a ring of P + 1 register sets, slot =
    // k-step mod (P + 1), the slice fully unrolled): one wave per SIMD -- N = 1024 has one 64x64 tile per CU -- has
    // nobody else to cover its LDS round trip, and the rolled `#pragma unroll 4` loop this replaces waited for both
    // ds_read_b128 of a k-step in front of its first FMA (ISA: s_waitcnt lgkmcnt(0) per k-step, ~157 cycles for 32 cycles
    // of v_pk_fma_f32; round 4).
    constexpr int SL = P + 1;
    f32x4 fa[SL][RI], fb[SL][RJ];
    auto request = [&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value, slot = kk % SL;
      const int g = swz_slot(kk >> 2);
#pragma unroll
      for (int h = 0; h < RI; ++h) fa[slot][h] = *reinterpret_cast<const f32x4 *>(As + kk * BM + 4 * ((ty + 16 * h) ^ g));
#pragma unroll
      for (int h = 0; h < RJ; ++h) fb[slot][h] = *reinterpret_cast<const f32x4 *>(Bs + kk * BN + 64 * h + 4 * tx);
    };
    static_for<P>([&](auto p_c) { request(p_c); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<KB>([&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value, slot = kk % SL;
      if constexpr (kk + P < KB) request(std::integral_constant<int, kk + P>{});
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const float a = fa[slot][i >> 2][i & 3];
#pragma unroll
        for (int j = 0; j < TJ / 2; ++j) {
          const f32x4 bv = fb[slot][j >> 1];
          const f32x2 b2 = (j & 1) ? f32x2{bv[2], bv[3]} : f32x2{bv[0], bv[1]};
          acc[i][j] = __builtin_elementwise_fma(f32x2{a, a}, b2, acc[i][j]);
        }
      }
      // pin the k-step: its FMAs done here, the next requests not before here (left alone, hipcc issues every read of
      // the unrolled slice first and the FMAs after them: 512 registers and spills)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ / 2; j += 2) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1])::"memory");
      __builtin_amdgcn_sched_barrier(0);
    });
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
        f32x4 v = {acc[i][2 * jh][0], acc[i][2 * jh][1], acc[i][2 * jh + 1][0], acc[i][2 * jh + 1][1]};
        *reinterpret_cast<f32x4 *>(C + (size_t)row * ldc + col) = v;
      } else if (row < m) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (col + u < n) C[(size_t)row * ldc + col + u] = acc[i][2 * jh + (u >> 1)][u & 1];
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
