// sgemm_mfma_small.hpp -- K2s: the MFMA SGEMM for problems too small to fill
// the chip with 128x128 tiles (N=1024 has only 64 of them for 256 CUs).
//
// Same arithmetic contract as K2 (sgemm_mfma.hpp): every C(i,j) is one fp32
// accumulator fed its k products in ascending k by v_mfma_f32_16x16x4_f32, so
// the result is bit-identical to every other kernel variant.  What changes is
// the granularity: a workgroup owns a 64x64 C tile, its 4 waves 32x32 each
// (2x2 MFMA tiles, 16 accumulator registers), so N=1024 yields 256 workgroups
// = 1024 waves, one per SIMD, and up to five workgroups fit a CU.
//
// Packing (same ideas as sgemm_tile.hpp, re-derived for 8-byte fragment reads):
//   As[k][m] k-major, 64 floats (256 B = one full bank row) per k-row; a wave
//   reads its two-tile fragment with ONE ds_read_b64 per operand: lane (i, kq)
//   takes As[k0+kq][m0+2i .. 2i+1] -> tile t covers rows {m0 + 2i + t}.
//   ds_read_b64 is served in two 32-lane halves over 64 banks; lanes with
//   kq = 0 and kq = 1 of one half would land on the same 32 banks one row
//   apart, so odd k-rows are stored with their two 32-float halves swapped
//   (slot bit 3 flipped) and the half reads 64 distinct banks.
//   The transposing A store (ds_write_b128 of a register-transposed 4x4 block)
//   XORs the low three slot bits with the k-chunk index, as in sgemm_tile.hpp.
//   Waves 0-1 stage A (one 4x4 block per thread), waves 2-3 stage B (four
//   float4 per thread): 4 loads + 4 LDS stores per thread per K-slice.
#pragma once
#include <type_traits>

#include "sgemm_tile.hpp"

namespace mmh {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool EDGE>
__global__ void __launch_bounds__(256)
sgemm_mfma_small_kernel(int m, int n, int k, const float *__restrict__ A, int lda,
                        const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                        int accumulate, int nbm, int nbn) {
  constexpr int BM = 64, BN = 64;
  constexpr int A_FLOATS = BK * BM, B_FLOATS = BK * BN, BUF = A_FLOATS + B_FLOATS;
  constexpr int KS = BK / 4;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];   // 32 KiB

  int tm, tn;
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, kq = lane >> 4;
  const bool stage_a = wave < 2;            // wave-uniform role

  const int rows_valid = EDGE ? min(BM, m - row0) : BM;
  const int cols_valid = EDGE ? min(BN, n - col0) : BN;
  const bool whole_c = !EDGE || (rows_valid == BM && cols_valid == BN);

  // C(t, u, r): row = crow + 2r + t, col = ccol + u
  const int crow = row0 + wm * 32 + 8 * kq;
  const int ccol = col0 + wn * 32 + 2 * li;
  typedef float c_vec_u __attribute__((ext_vector_type(2), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, f32x2>;

  f32x4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 2 * r + t;
      f32x2 v = {0.f, 0.f};
      if (accumulate) {
        if (whole_c) {
          v = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + ccol);
        } else if (row < m) {
          if (ccol < n) v[0] = C[(size_t)row * ldc + ccol];
          if (ccol + 1 < n) v[1] = C[(size_t)row * ldc + ccol + 1];
        }
      }
      acc[t][0][r] = v[0];
      acc[t][1][r] = v[1];
    }

  // ---- staging: buffer descriptors bound every read (see sgemm_mfma.hpp) ----
  const uint32_t ext_a = EDGE ? (uint32_t)(((rows_valid - 1) * lda + k) * 4) : 0x7fffffffu;
  const uint32_t ext_b = EDGE ? (uint32_t)(((k - 1) * ldb + cols_valid) * 4) : 0x7fffffffu;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A + (size_t)row0 * lda), 0, ext_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B + col0), 0, ext_b, 0x00020000);
  const int t2 = tid & 127;
  const int sa_c = t2 & 7, sa_q = t2 >> 3;          // A: k-chunk, row block
  const int sb_cb = t2 & 15, sb_kr = t2 >> 4;       // B: column slot, first k-row
  const uint32_t voff = stage_a ? (uint32_t)((4 * sa_q) * lda + 4 * sa_c) * 4u
                                : (uint32_t)(sb_kr * ldb + 4 * sb_cb) * 4u;
  const int nk = (k + BK - 1) / BK;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  f32x4 sr[4];
  auto stage_load = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int soff = stage_a ? (k0 + j * lda) * 4 : (k0 + 8 * j) * ldb * 4;
      sr[j] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(stage_a ? rsrc_a : rsrc_b, voff, soff, 0));
    }
    if (EDGE && stage_a && kt == nk - 1 && (k % BK) != 0) {
      const int krem = k - k0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (4 * sa_c + s >= krem) sr[j][s] = 0.0f;
    }
  };
  auto stage_store = [&](float *buf) {
    if (stage_a) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int slot = sa_q ^ (sa_c & 7) ^ ((s & 1) << 3);
        f32x4 v = {sr[0][s], sr[1][s], sr[2][s], sr[3][s]};
        *reinterpret_cast<f32x4 *>(buf + (4 * sa_c + s) * BM + 4 * slot) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kr = sb_kr + 8 * j;
        *reinterpret_cast<f32x4 *>(buf + A_FLOATS + kr * BN + 4 * (sb_cb ^ ((kr & 1) << 3))) = sr[j];
      }
    }
  };
  // fragment reads: float offsets inside a buffer
  auto frag_a = [&](const float *buf, int ks) {
    const int slot = (wm * 8 + (li >> 1)) ^ (ks & 7) ^ ((kq & 1) << 3);
    return *reinterpret_cast<const f32x2 *>(buf + (4 * ks + kq) * BM + 4 * slot + 2 * (li & 1));
  };
  auto frag_b = [&](const float *buf, int ks) {
    const int slot = (wn * 8 + (li >> 1)) ^ ((kq & 1) << 3);
    return *reinterpret_cast<const f32x2 *>(buf + A_FLOATS + (4 * ks + kq) * BN + 4 * slot +
                                            2 * (li & 1));
  };

  f32x2 fa[2], fb[2];
  if (nk > 0) {
    stage_load(0);
    stage_store(lds);
    if (nk > 1) stage_load(1);
  }
  __syncthreads();
  if (nk > 0) {
    fa[0] = frag_a(lds, 0);
    fb[0] = frag_b(lds, 0);
  }
  int cur = 0;
  auto slice = [&](int kt, auto more_c, auto more2_c) {
    constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
    const float *buf = lds + cur * BUF;
    float *nxt = lds + (cur ^ 1) * BUF;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
        fa[(ks + 1) & 1] = frag_a(buf, ks + 1);
        fb[(ks + 1) & 1] = frag_b(buf, ks + 1);
      } else {
        // slice hand-over pipelined across the barrier, as in K2
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (MORE) {
          fa[(ks + 1) & 1] = frag_a(nxt, 0);
          fb[(ks + 1) & 1] = frag_b(nxt, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ks == 1 && MORE) stage_store(nxt);
      if (ks == 3 && MORE2) stage_load(kt + 2);
      const f32x2 a = fa[ks & 1], b = fb[ks & 1];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
    cur ^= 1;
  };
  using T = std::true_type;
  using F = std::false_type;
  int kt = 0;
  for (; kt + 2 < nk; ++kt) slice(kt, T{}, T{});
  if (kt + 1 < nk) { slice(kt, T{}, F{}); ++kt; }
  if (kt < nk) slice(kt, F{}, F{});

#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 2 * r + t;
      f32x2 v = {acc[t][0][r], acc[t][1][r]};
      if (whole_c) {
        *reinterpret_cast<c_vec *>(C + (size_t)row * ldc + ccol) = v;
      } else if (row < m) {
        if (ccol < n) C[(size_t)row * ldc + ccol] = v[0];
        if (ccol + 1 < n) C[(size_t)row * ldc + ccol + 1] = v[1];
      }
    }
}

}  // namespace mmh
