// igemm_s8_k3.hpp (tools build only, -DMMH_AB_BUILD) -- the int8 rungs the product no longer ships (round 6: MMH_OPT_IGEMM_MODE 0
// never reached them once the in-place kernels took every 4-byte aligned operand and the workspace copy the rest):
//   K3   igemm_s8_kernel: 128x128 tile, B transposed inside the kernel with v_perm_b32 (mode 1);
//   K3d  pack_bt_s8_kernel + igemm_s8_dma_kernel<..., BTR = false>: B packed once per call (modes 3 / 4), and the timing-only
//        ablations of the packed 256x256 kernel (modes 10..13: WRONG results).
// The descriptions of both are in csrc/igemm_s8.hpp's header, where they were written.
#pragma once
#include "igemm_s8.hpp"

namespace mmh {

// --------------------------------------------------------------------------
// K3
// --------------------------------------------------------------------------

// 4x4 byte transpose: rows r0..r3 (4 bytes each) -> columns c0..c3
__device__ __forceinline__ void transpose4x4_bytes(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3,
                                                   uint32_t (&c)[4]) {
  // v_perm_b32(hi, lo, sel): result byte i = byte sel[i] of the 8-byte {hi:lo}
  const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400);  // r0.0 r1.0 r0.1 r1.1
  const uint32_t t1 = __builtin_amdgcn_perm(r3, r2, 0x05010400);  // r2.0 r3.0 r2.1 r3.1
  const uint32_t t2 = __builtin_amdgcn_perm(r1, r0, 0x07030602);  // r0.2 r1.2 r0.3 r1.3
  const uint32_t t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602);  // r2.2 r3.2 r2.3 r3.3
  c[0] = __builtin_amdgcn_perm(t1, t0, 0x05040100);               // r0.0 r1.0 r2.0 r3.0
  c[1] = __builtin_amdgcn_perm(t1, t0, 0x07060302);               // r0.1 r1.1 r2.1 r3.1
  c[2] = __builtin_amdgcn_perm(t3, t2, 0x05040100);
  c[3] = __builtin_amdgcn_perm(t3, t2, 0x07060302);
}

template <bool EDGE>
__global__ void __launch_bounds__(256, 2)
igemm_s8_kernel(int m, int n, int k, const int8_t *__restrict__ A, int lda,
                const int8_t *__restrict__ B, int ldb, int32_t *__restrict__ C, int ldc,
                int accumulate, int nbm, int nbn) {
  constexpr int BM = 128, BN = 128;
  extern __shared__ __attribute__((aligned(16))) int8_t ilds[];   // 2 x (A image | B image) = 64 KiB

  const int tile = blockIdx.x;
  const int tm = tile / nbn, tn = tile % nbn;
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, g = lane >> 4;

  const int rows_valid = EDGE ? min(BM, m - row0) : BM;
  const int cols_valid = EDGE ? min(BN, n - col0) : BN;
  const bool whole_c = !EDGE || (rows_valid == BM && cols_valid == BN);
  const int crow = row0 + wm * 64 + 4 * g;     // + 16 t + r
  const int ccol = col0 + wn * 64 + 4 * li;    // .. +3 (u)
  typedef int c_vec_u __attribute__((ext_vector_type(4), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, i32x4>;

  i32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 16 * t + r;
      i32x4 v = {0, 0, 0, 0};
      if (accumulate) {
        if (whole_c) {
          v = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + ccol);
        } else if (row < m) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (ccol + u < n) v[u] = C[(size_t)row * ldc + ccol + u];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u][r] = v[u];
    }

  // ---- staging (per thread: 4 x 16 B of A, 4 x 16 B of B per slice) ----
  // extents in bytes, rounded up to whole dwords: the range check is per DWORD, so a
  // byte-exact extent would zero the last valid bytes of the last row (lda, ldb are
  // multiples of 4 here, so the round-up stays inside the row)
  const uint32_t ext_a = (uint32_t)((rows_valid - 1) * lda + ((k + 3) & ~3));
  const uint32_t ext_b = (uint32_t)((k - 1) * ldb + ((cols_valid + 3) & ~3));
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(A + (size_t)row0 * lda), 0, ext_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(B + col0), 0, ext_b, 0x00020000);
  const int a_row = tid >> 3, a_ch = tid & 7;        // A: rows a_row + 32 p, 16-byte chunk a_ch
  const int b_nb = tid & 7, b_kb = tid >> 3;         // B: columns 16 b_nb.., k rows 4 b_kb..+3
  const uint32_t voff_a = (uint32_t)(a_row * lda + 16 * a_ch);
  const uint32_t voff_b = (uint32_t)(4 * b_kb * ldb + 16 * b_nb);
  i32x4 sa[4], sb[4];
  auto stage_load = [&](int kt) {
    const int k0 = kt * IK;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      sa[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a, k0 + 32 * p * lda, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      sb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, voff_b, (k0 + j) * ldb, 0);
  };
  auto stage_store = [&](int8_t *buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = a_row + 32 * p;
      *reinterpret_cast<i32x4 *>(buf + row * IK + 16 * (a_ch ^ ((row >> 1) & 7))) = sa[p];
    }
    // B: dword q of the four loaded k-rows holds columns 16 b_nb + 4q .. +3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t col[4];
      transpose4x4_bytes((uint32_t)sb[0][q], (uint32_t)sb[1][q], (uint32_t)sb[2][q], (uint32_t)sb[3][q], col);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int nloc = 16 * b_nb + 4 * q + c;                 // column inside the tile
        const int prow = (nloc & 3) * 32 + (nloc >> 2);          // u-major row of the B image
        const int slot = (b_kb >> 2) ^ ((prow >> 1) & 7);
        *reinterpret_cast<uint32_t *>(buf + ITILE + prow * IK + 16 * slot + 4 * (b_kb & 3)) = col[c];
      }
    }
  };
  // fragment reads for MFMA step s (0/1) of a slice
  const int swz = (li >> 1) & 7;
  auto frag_a = [&](const int8_t *buf, int s, int t) {
    const int row = wm * 64 + 16 * t + li;
    return *reinterpret_cast<const i32x4 *>(buf + row * IK + 16 * ((4 * s + g) ^ swz));
  };
  auto frag_b = [&](const int8_t *buf, int s, int u) {
    const int prow = u * 32 + wn * 16 + li;
    return *reinterpret_cast<const i32x4 *>(buf + ITILE + prow * IK + 16 * ((4 * s + g) ^ swz));
  };

  const int nk = (k + IK - 1) / IK;
  i32x4 fa[2][4], fb[2][4];
  if (nk > 0) {
    stage_load(0);
    stage_store(ilds);
    if (nk > 1) stage_load(1);
  }
  __syncthreads();
  if (nk > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[0][t] = frag_a(ilds, 0, t); fb[0][t] = frag_b(ilds, 0, t); }
  }
  int cur = 0;
  auto slice = [&](int kt, auto more_c, auto more2_c) {
    constexpr bool MORE = decltype(more_c)::value, MORE2 = decltype(more2_c)::value;
    const int8_t *buf = ilds + cur * 2 * ITILE;
    int8_t *nxt = ilds + (cur ^ 1) * 2 * ITILE;
    // step 0: prefetch step 1's fragments, write the next slice to LDS, MFMAs
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[1][t] = frag_a(buf, 1, t); fb[1][t] = frag_b(buf, 1, t); }
    if (MORE) stage_store(nxt);
    if (MORE2) stage_load(kt + 2);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[0][t], fb[0][u], acc[t][u], 0, 0, 0);
    // Pipeline description for the scheduler (as in K2): the 8 fragment reads of
    // step 1 up front, then the 20 LDS stores and 8 global loads dealt out between
    // the 16 MFMAs instead of in bursts.
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);             // DS read
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // MFMA
      if (MORE && i < 10) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);               // DS write
      if (MORE2 && i >= 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // VMEM read
    }
    // hand-over: step 1's fragments are in registers; barrier; prefetch the next
    // slice's step 0; step 1's MFMAs cover that LDS latency
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (MORE) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { fa[0][t] = frag_a(nxt, 0, t); fb[0][t] = frag_b(nxt, 0, t); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[1][t], fb[1][u], acc[t][u], 0, 0, 0);
    cur ^= 1;
  };
  using T = std::true_type;
  using F = std::false_type;
  int kt = 0;
  for (; kt + 2 < nk; ++kt) slice(kt, T{}, T{});
  if (kt + 1 < nk) { slice(kt, T{}, F{}); ++kt; }
  if (kt < nk) slice(kt, F{}, F{});

#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 16 * t + r;
      i32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
      if (whole_c) {
        *reinterpret_cast<c_vec *>(C + (size_t)row * ldc + ccol) = v;
      } else if (row < m) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ccol + u < n) C[(size_t)row * ldc + ccol + u] = v[u];
      }
    }
}

// Bt[(n_pad)][kp] <- transpose of B[k][n] (ldb), zero padded.  One thread moves a
// 16(k) x 4(n) byte block: 16 dword loads, four 4x4 byte transposes, four 16-byte stores.
__global__ void __launch_bounds__(256) pack_bt_s8_kernel(const int8_t *__restrict__ B, int ldb, int k,
                                                         int n, int8_t *__restrict__ Bt, int kp,
                                                         int n_pad, int b_dword_ok) {
  const int nq = threadIdx.x & 15, kq = threadIdx.x >> 4;
  const int n0 = blockIdx.x * 64 + 4 * nq, k0 = blockIdx.y * 256 + 16 * kq;
  if (n0 >= n_pad || k0 >= kp) return;
  uint32_t rows[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kk = k0 + j;
    uint32_t w = 0;
    if (kk < k) {
      if (b_dword_ok && n0 + 3 < n) {
        w = *reinterpret_cast<const uint32_t *>(B + (size_t)kk * ldb + n0);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (n0 + c < n) w |= (uint32_t)(uint8_t)B[(size_t)kk * ldb + n0 + c] << (8 * c);
      }
    }
    rows[j] = w;
  }
  uint32_t col[4][4];   // col[c][g] = k bytes 4g..4g+3 of column n0 + c
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint32_t t[4];
    transpose4x4_bytes(rows[4 * g], rows[4 * g + 1], rows[4 * g + 2], rows[4 * g + 3], t);
#pragma unroll
    for (int c = 0; c < 4; ++c) col[c][g] = t[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    i32x4 v = {(int)col[c][0], (int)col[c][1], (int)col[c][2], (int)col[c][3]};
    *reinterpret_cast<i32x4 *>(Bt + (size_t)(n0 + c) * kp + k0) = v;
  }
}

// Workspace bytes mmh_igemm_s8 needs for the packed B of a (k x n) problem.
inline size_t igemm_s8_pack_bytes(int n, int k) {
  const size_t n_pad = ((size_t)n + 127) & ~(size_t)127, kp = ((size_t)k + 255) & ~(size_t)255;
  return n_pad * kp;
}

// mode: 0 = K3d when eligible (needs `bt_ws`, >= igemm_s8_pack_bytes; 256x256 tiles when there
//           is at least one per CU, else 128x128), else K3 / simple;
//       1 = K3 (in-kernel transpose), 2 = the simple kernel,
//       3 / 4 = K3d with 128x128 / 256x256 tiles forced (A/B switch).
// does this call need the packed-B workspace (igemm_s8_pack_bytes)?
inline bool igemm_s8_needs_pack(int mode, const int8_t *A, int lda, const int8_t *B, int ldb, int k) {
  (void)A; (void)lda; (void)B; (void)ldb; (void)k;
  return mode == 3 || mode == 4 || mode >= 10;
}


// modes 1, 3, 4, 10..13; anything else (or an operand these kernels cannot take): hipErrorNotSupported
inline hipError_t launch_igemm_s8_ab(int m, int n, int k, const int8_t *A, int lda, const int8_t *B, int ldb, int32_t *C, int ldc,
                                     int acc, hipStream_t s, int8_t *bt_ws, int mode) {
  const int nbm = (m + 127) / 128, nbn = (n + 127) / 128;
  dim3 grid((unsigned)(nbm * nbn)), block(256);
  const bool shape_ok = (m % 128 == 0) && (n % 128 == 0);
  const bool a4 = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 3) == 0);
  const bool b4 = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 3) == 0);
  const size_t lim = (1ull << 31) - 4096;
  const size_t kp = ((size_t)k + 255) & ~(size_t)255, n_pad = ((size_t)n + 127) & ~(size_t)127;
  constexpr size_t lds = 4 * ITILE;   // 64 KiB
  const bool c_fast = shape_ok && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  if ((mode == 3 || mode == 4 || mode >= 10) && bt_ws && a4 && ((size_t)256 * lda + k) < lim && 256 * kp < lim) {
    dim3 pgrid((unsigned)(n_pad / 64), (unsigned)((kp + 255) / 256));
    hipLaunchKernelGGL(pack_bt_s8_kernel, pgrid, dim3(256), 0, s, B, ldb, k, n, bt_ws, (int)kp, (int)n_pad, b4 ? 1 : 0);
    const int kpi = (int)kp, npi = (int)n_pad;
    const bool whole256 = (m % 256 == 0) && (n % 256 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    if (mode >= 10 && !whole256) return hipErrorInvalidValue;
    if (mode == 10) return launch_igemm_s8_dma_edge<256, 256, 8, false, 1>(m, n, k, A, lda, bt_ws, kpi, npi, C, ldc, acc, s);
    if (mode == 11) return launch_igemm_s8_dma_edge<256, 256, 8, false, 2>(m, n, k, A, lda, bt_ws, kpi, npi, C, ldc, acc, s);
    if (mode == 12) return launch_igemm_s8_dma_edge<256, 256, 8, false, 3>(m, n, k, A, lda, bt_ws, kpi, npi, C, ldc, acc, s);
    if (mode == 13) return launch_igemm_s8_dma_edge<256, 256, 8, false, 4>(m, n, k, A, lda, bt_ws, kpi, npi, C, ldc, acc, s);
    if (mode == 3) return launch_igemm_s8_dma<128, 128, 4>(m, n, k, A, lda, bt_ws, kpi, npi, C, ldc, acc, s);
    return launch_igemm_s8_dma<256, 256, 8>(m, n, k, A, lda, bt_ws, kpi, npi, C, ldc, acc, s);
  }
  const bool window_ok = ((size_t)128 * lda + k) < lim && ((size_t)k * ldb + 128) < lim;
  if (mode == 1 && a4 && b4 && window_ok) {
    if (c_fast) hipLaunchKernelGGL(igemm_s8_kernel<false>, grid, block, lds, s, m, n, k, A, lda, B, ldb, C, ldc, acc, nbm, nbn);
    else hipLaunchKernelGGL(igemm_s8_kernel<true>, grid, block, lds, s, m, n, k, A, lda, B, ldb, C, ldc, acc, nbm, nbn);
    return hipGetLastError();
  }
  return hipErrorNotSupported;
}

}  // namespace mmh
