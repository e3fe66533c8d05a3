// igemm_s8_pp.hpp -- K3p: the int8 GEMM's 256x256 tile with its two wave groups in PING-PONG.
//
// Same arithmetic, same LDS images and the same fragment reads as K3t (igemm_s8.hpp: B read in place by
// LDS-DMA, fragments by ds_read_b64_tr_b8, v_mfma_i32_16x16x64_i8, exact integers) -- what changes is WHEN
// each wave does what.  In K3t all eight waves run the same phase at once: each SIMD's two waves both want
// the matrix pipe, both interleave their fragment reads with their MFMAs, a whole slice of LDS-DMA (64 pieces
// per workgroup) is requested in one phase and drained with `vmcnt(0)` once per slice; the matrix pipe is busy
// 45-61 % of the launch (profiles/r02_igemm_s8_rocprofv3.json).  Here (cdna guide, "8-phase" schedule):
//   * the workgroup's two wave groups -- waves 0-3 (C rows 0-127) and waves 4-7 (rows 128-255), one wave of
//     each per SIMD -- run ONE BARRIER apart.  A phase of a wave is  [R: fragment reads for 16 MFMAs, two
//     LDS-DMA pieces] barrier [M: the 16 MFMAs, s_setprio 1] barrier , so while one wave of a SIMD sits in M
//     the other is in R: the pipe always has a wave that does nothing but MFMAs, the LDS a wave that does
//     nothing but reads;
//   * the LDS-DMA of a slice is dealt out over the phases, two pieces per wave and phase, each region of the
//     double buffer re-requested as soon as its last reader is done, FIVE phases ahead of its first reader;
//     the wait is a counted `vmcnt(6)` per phase -- never 0 in the loop -- one phase and one barrier ahead of
//     the reads it guards.
// Regions of a slice's image: A0 / A1 = rows 0-127 / 128-255 of the A image (read by group 0 / group 1 only),
// B0 / B1 = k rows 0-63 / 64-127 of the B image (MFMA step 0 / 1).  Phases of slice t: (step, half) =
// (0,0) (0,1) (1,0) (1,1), B fragments read in the half-0 phases and kept for half 1.  Request schedule --
// phase 0: A1 of slice t+1, 1: B1 of t+1, 2: B0 of t+2, 3: A0 of t+2 -- each into the buffer region whose
// readers finished >= 1 phase (and one lgkmcnt(0) + barrier) earlier.
// Measured (profiles/r03_igemm_s8_ksweep.txt, r03_notes.md section 4; M = N = 4096, K swept, us per launch = fixed + slope x K): K3t 16.8 us + 3.05 POPS
// in the loop; this kernel with 16 MFMAs per phase 3.1, with 32 MFMAs per phase (PPS = 2, what ships) 17.1 us +
// 3.18 POPS = 0.83 of what the matrix pipe sustains on random operands at the power-managed clock (3.84 POPS).  The
// fixed part -- launch, prologue and above all the 64 MB C store, which nothing can overlap with one tile per CU
// and every accumulator register taken -- is what keeps 4096^3 at 2.3 POPS.
// BASELINE.json config 5; no reference code (README.md:71-85 is prose): parity unpinned.
#pragma once
#include "igemm_s8.hpp"

namespace mmh {

// Two transposing LDS reads (8 + 8 consecutive k of one column each; lo = k .. k+7, hi = k+8 .. k+15 of a 64-deep
// step) from one address register.  Inline asm, see igemm_s8_pp_kernel; a free function because an asm statement
// whose operands are captured by a lambda inside a __global__ template does not survive hipcc's host pass.
typedef int pp_i32x2 __attribute__((ext_vector_type(2)));
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ void ds_read_tr8_pair(uint32_t addr, pp_i32x2 &lo, pp_i32x2 &hi) {
  asm volatile("ds_read_b64_tr_b8 %0, %2 offset:%3\n\tds_read_b64_tr_b8 %1, %2 offset:%4"
               : "=v"(lo), "=v"(hi)
               : "v"(addr), "n"(OFF_LO), "n"(OFF_HI));
}

// PPS = phases per slice: 4 (16 MFMAs between barriers: one 64-row half of the wave tile per 64-deep step) or
// 2 (32 MFMAs: the whole wave tile per step -- half the barriers per MFMA; the request schedule for it is in `phase2`).
template <bool EDGE, bool DEQ, int PPS>
__global__ void __launch_bounds__(512, 1)
igemm_s8_pp_kernel(int m, int n, int k, const int8_t *__restrict__ A, int lda, const int8_t *__restrict__ B, int ldb,
                   int32_t *__restrict__ C, int ldc, int accumulate, int nbm, int nbn, const float *__restrict__ deq) {
  constexpr int BM = 256, BN = 256, TM = 8, TN = 4;
  constexpr int A_IMG = BM * IK, B_IMG = BN * IK, STAGE = A_IMG + B_IMG;   // 32 KiB + 32 KiB
  extern __shared__ __attribute__((aligned(16))) int8_t ilds[];            // 2 x STAGE = 128 KiB

  int tm, tn;
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;     // wm = the wave's group (0: older half, 1: younger half)
  const int li = lane & 15, g = lane >> 4;

  const int rows_valid = EDGE ? min(BM, m - row0) : BM;
  const bool whole_c = !EDGE || (rows_valid == BM && col0 + BN <= n);
  // lane (li, g) holds C[crow + 16 t][ccol + 16 u + r] (the MFMA operands are swapped: D = tile^T)
  const int crow = row0 + wm * 128 + li;
  const int ccol = col0 + wn * 64 + 4 * g;
  typedef int c_vec_u __attribute__((ext_vector_type(4), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, i32x4>;

  i32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) acc[t][u] = i32x4{0, 0, 0, 0};
  if (accumulate) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int row = crow + 16 * t, col = ccol + 16 * u;
        if (whole_c) {
          acc[t][u] = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + col);
        } else if (row < m) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < n) acc[t][u][e] = C[(size_t)row * ldc + col + e];
        }
      }
  }

  const int nk = 2 * ((k + 2 * IK - 1) / (2 * IK));   // slices, rounded up to even (k > 0)
  // descriptors as in K3t: A bounded at the block's last valid row, B (row-major, in place) at row k
  const uint32_t ext_a = (uint32_t)((rows_valid - 1) * lda + ((k + 3) & ~3));
  const uint32_t ext_b = (uint32_t)((k - 1) * ldb + ((min(BN, n - col0) + 3) & ~3));
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(A + (size_t)row0 * lda), 0, ext_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(B + col0), 0, ext_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(A), 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(B), 0, 0, 0x00020000);
  auto btr_swz = [](int r) { return (r & 7) | (((r >> 4) & 1) << 3); };
  // Each region is 16 pieces of 1 KiB; wave w moves pieces 2 w and 2 w + 1 of whatever region is requested.
  //   A region h: piece j = image rows 128 h + 8 j .. + 7 (128 B each), lane -> row lane / 8, 16-byte slot lane % 8
  //   B region s: piece j = k rows 64 s + 4 j .. + 3 (256 B each),   lane -> row lane / 16, slot lane % 16
  uint32_t voff_a[2][2], voff_b[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int prow = 128 * h + 8 * (2 * wave + j) + (lane >> 3);
      voff_a[h][j] = (uint32_t)(prow * lda + 16 * ((lane & 7) ^ ((prow >> 1) & 7)));
      const int r = 64 * h + 4 * (2 * wave + j) + (lane >> 4);
      voff_b[h][j] = (uint32_t)(r * ldb + 16 * ((lane & 15) ^ btr_swz(r)));
    }
  // request region `reg` (0: A0, 1: A1, 2: B0, 3: B1) of slice kt into buffer `buf`
  auto request = [&](auto reg_c, int8_t *buf, int kt) {
    constexpr int REG = decltype(reg_c)::value, HS = REG & 1;
    const bool live = kt < nk;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (REG < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? rsrc_a : null_a,
                                                 (__attribute__((address_space(3))) void *)(buf + (128 * HS + 8 * (2 * wave + j)) * IK),
                                                 16, voff_a[HS][j], kt * IK, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? rsrc_b : null_b,
                                                 (__attribute__((address_space(3))) void *)(buf + A_IMG + (64 * HS + 4 * (2 * wave + j)) * BN),
                                                 16, voff_b[HS][j], kt * IK * ldb, 0, 0);
    }
  };
  // fragment addresses (K3t's): A image [row][128 B] with the slot XOR, B image [k row][256 B] with btr_swz
  const int swz = (li >> 1) & 7;
  uint32_t a_off[2][2], bt_off[2][TN];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int st = 0; st < 2; ++st) a_off[c][st] = (uint32_t)(c * STAGE + (wm * 128 + li) * IK + 16 * ((4 * st + g) ^ swz));
    const int r = 16 * g + (li >> 1);
#pragma unroll
    for (int u = 0; u < TN; ++u)
      bt_off[c][u] = (uint32_t)(c * STAGE + A_IMG + r * BN + 16 * ((4 * wn + u) ^ btr_swz(r)) + 8 * (li & 1));
  }
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x4 fa[PPS == 2 ? 8 : 4], fb[TN];
  auto read_a = [&](auto c_c, auto st_c, auto h_c) {   // A tiles 4 h .. 4 h + 3 of step st, buffer c (PPS 2: all eight)
    constexpr int CB_ = decltype(c_c)::value, ST = decltype(st_c)::value, H = decltype(h_c)::value;
#pragma unroll
    for (int t = 0; t < (PPS == 2 ? 8 : 4); ++t)
      fa[t] = *reinterpret_cast<const i32x4 *>(ilds + a_off[CB_][ST] + 16 * (4 * H + t) * IK);
  };
  // The transposing reads are spelled in inline asm: through the builtin hipcc cannot tell what the read may alias
  // and puts `s_waitcnt vmcnt(0)` in front of it whenever an LDS-DMA is in flight -- which here is always, by design.
  // (cdna guide 5.7, form (iii): "=v" loads, a wait-only statement, sched_barrier(0) before the first consumer --
  // all three sit in `phase` below, in front of the barrier that precedes the MFMAs.)
  const uint32_t lds_base = (uint32_t)(uintptr_t)ilds;
  i32x2 fb_lo[TN], fb_hi[TN];
  auto read_b = [&](auto c_c, auto st_c) {             // the B tiles of step st, buffer c
    constexpr int CB_ = decltype(c_c)::value, ST = decltype(st_c)::value;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      ds_read_tr8_pair<64 * ST * BN, (64 * ST + 8) * BN>(lds_base + bt_off[CB_][u], fb_lo[u], fb_hi[u]);
    }
  };
  constexpr std::integral_constant<int, 0> i0{};
  constexpr std::integral_constant<int, 1> i1{};
  constexpr std::integral_constant<int, 2> i2{};
  constexpr std::integral_constant<int, 3> i3{};

  if constexpr (PPS == 4) {
  // ---- prologue: slice 0 whole, B0 and A0 of slice 1 -- six request groups, the two oldest landed ----
  request(i2, ilds, 0);
  request(i0, ilds, 0);
  request(i1, ilds, 0);
  request(i3, ilds, 0);
  request(i2, ilds + STAGE, 1);
  request(i0, ilds + STAGE, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // the younger group runs one barrier behind from here on

  // One phase: R (wait for what the NEXT phase's reads need, read this phase's fragments, request two pieces),
  // barrier, M (16 MFMAs), barrier.
  auto phase = [&](int kt, auto cur_c, auto p_c) {
    constexpr int CUR = decltype(cur_c)::value, P = decltype(p_c)::value, S = P >> 1, H = P & 1;
    constexpr std::integral_constant<int, CUR> cur{};
    int8_t *mine = ilds + CUR * STAGE, *other = ilds + (CUR ^ 1) * STAGE;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    read_a(cur, std::integral_constant<int, S>{}, std::integral_constant<int, H>{});
    if constexpr (H == 0) read_b(cur, std::integral_constant<int, S>{});
    if constexpr (P == 0) request(i1, other, kt + 1);
    if constexpr (P == 1) request(i3, other, kt + 1);
    if constexpr (P == 2) request(i2, mine, kt + 2);
    if constexpr (P == 3) request(i0, mine, kt + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (H == 0) {
#pragma unroll
      for (int u = 0; u < TN; ++u) fb[u] = i32x4{fb_lo[u][0], fb_lo[u][1], fb_hi[u][0], fb_hi[u][1]};
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u)
        acc[4 * H + t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[u], fa[t], acc[4 * H + t][u], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    phase(kt, i0, i0);
    phase(kt, i0, i1);
    phase(kt, i0, i2);
    phase(kt, i0, i3);
    phase(kt + 1, i1, i0);
    phase(kt + 1, i1, i1);
    phase(kt + 1, i1, i2);
    phase(kt + 1, i1, i3);
  }
  } else {
  // ---- PPS == 2.  Request schedule, four pieces per wave and phase, issued at the TOP of the phase's R section:
  //   phase 0 of slice t: A0 and A1 of slice t+1 (other buffer; its A regions were last read in slice t-1)
  //   phase 1 of slice t: B1 of slice t+1 (other buffer), B0 of slice t+2 (this buffer, last read in phase 0)
  // every region two or three phases ahead of its first reader; the wait, after the phase's own reads, is
  // `vmcnt(4)`: everything but the four pieces just requested -- which is what the NEXT phase's reads need.
  request(i2, ilds, 0);
  request(i0, ilds, 0);
  request(i1, ilds, 0);
  request(i3, ilds, 0);
  request(i2, ilds + STAGE, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  auto phase2 = [&](int kt, auto cur_c, auto s_c) {
    constexpr int CUR = decltype(cur_c)::value, S = decltype(s_c)::value;
    constexpr std::integral_constant<int, CUR> cur{};
    int8_t *mine = ilds + CUR * STAGE, *other = ilds + (CUR ^ 1) * STAGE;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) {
      request(i0, other, kt + 1);
      request(i1, other, kt + 1);
    } else {
      request(i3, other, kt + 1);
      request(i2, mine, kt + 2);
    }
    read_a(cur, std::integral_constant<int, S>{}, i0);
    read_b(cur, std::integral_constant<int, S>{});
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < TN; ++u) fb[u] = i32x4{fb_lo[u][0], fb_lo[u][1], fb_hi[u][0], fb_hi[u][1]};
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[u], fa[t], acc[t][u], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    phase2(kt, i0, i0);
    phase2(kt, i0, i1);
    phase2(kt + 1, i1, i0);
    phase2(kt + 1, i1, i1);
  }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // the older group's matching barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-length tail requests

  const float deq_inv = DEQ ? 1.0f / (deq[0] * deq[1]) : 0.0f;
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int row = crow + 16 * t, col = ccol + 16 * u;
      i32x4 v = acc[t][u];
      if constexpr (DEQ) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t f = {(float)v[0] * deq_inv, (float)v[1] * deq_inv, (float)v[2] * deq_inv, (float)v[3] * deq_inv};
        v = __builtin_bit_cast(i32x4, f);
      }
      if (whole_c) {
        // (non-temporal stores were measured here: the fixed part of a 4096^3 launch grows from 17.1 to 21.2 us)
        *reinterpret_cast<c_vec *>(C + (size_t)row * ldc + col) = v;
      } else if (row < m) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < n) C[(size_t)row * ldc + col + e] = v[e];
      }
    }
}

template <int PPS>
inline hipError_t launch_igemm_s8_pp(int m, int n, int k, const int8_t *A, int lda, const int8_t *B, int ldb, int32_t *C,
                                     int ldc, int acc, hipStream_t s, const float *deq = nullptr) {
  constexpr int BM = 256, BN = 256;
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * IK;
  const bool c_fast = (m % BM == 0) && (n % BN == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#define MMH_PP_LAUNCH(E, D)                                                                                           \
  do {                                                                                                                \
    const hipError_t e = opt_in_big_lds(reinterpret_cast<const void *>(&igemm_s8_pp_kernel<E, D, PPS>), lds);              \
    if (e != hipSuccess) return e;                                                                                    \
    hipLaunchKernelGGL((igemm_s8_pp_kernel<E, D, PPS>), dim3((unsigned)(nbm * nbn)), dim3(512), lds, s, m, n, k, A, lda, B, \
                       ldb, C, ldc, acc, nbm, nbn, deq);                                                              \
  } while (0)
  if (deq) {
    if (c_fast) MMH_PP_LAUNCH(false, true);
    else MMH_PP_LAUNCH(true, true);
  } else {
    if (c_fast) MMH_PP_LAUNCH(false, false);
    else MMH_PP_LAUNCH(true, false);
  }
#undef MMH_PP_LAUNCH
  return hipGetLastError();
}

}  // namespace mmh
