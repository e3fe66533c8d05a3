// igemm_s8.hpp -- K3*: int8 x int8 -> int32 GEMM on v_mfma_i32_16x16x64_i8
// (BASELINE.json config 5).
//
// The reference tree has NO int8 code (aarch64-int8/ is an empty submodule;
// README.md:71-85 describes chgemm in prose: symmetric quantisation, inputs in
// [-127,127], int32 accumulation).  The contract implemented here is that
// prose plus armv7/REF_MMult.c:9-22's loop with the types changed:
//   C[m x n] (int32, row-major) = A[m x k] (int8) * B[k x n] (int8) (+ C).
// Integer arithmetic is exact and order-independent, so the result is
// bit-equal to an int32 triple loop for any k <= ~133000 (the parity tests check
// exactly that).
//
// gfx950 offers 16x16x32 and the double-rate 16x16x64 i8 MFMA; the 64-deep
// form is used: per lane 16 consecutive-k bytes of one A row
// and of one B column.  Because both operands use the same lane->k mapping,
// correctness does not depend on how the hardware numbers the k's inside.
//
// The kernels (K3 and K3d's packing pass moved to tools/ab/igemm_s8_k3.hpp in round 6; their descriptions stay for the
// tools build), in the order the launchers prefer them:
//  * K3t  igemm_s8_dma_kernel<..., BTR = true>: B read in place -- both operands global -> LDS by
//    LDS-DMA, B's row-major slices gathered into MFMA fragments by ds_read_b64_tr_b8 (below);
//  * K3d  pack_bt_s8_kernel + igemm_s8_dma_kernel<..., BTR = false>: B packed once per call into
//    Bt[n][k], for operands whose B is not dword-aligned;
//  * K3   igemm_s8_kernel: 128x128 tile, 4 waves x (4x4 MFMA tiles), K-slices
//    of 128 bytes double-buffered in LDS, global loads one slice ahead in
//    registers, slice hand-over pipelined across the barrier (as K2 does for
//    fp32).  A rows are k-contiguous in memory and go to LDS as they are; B is
//    n-contiguous, so each thread transposes 4(k) x 16(n) bytes with v_perm_b32
//    and stores one dword (4 k's) per column into Bt[n][k].  Both LDS images
//    are [row][128 B] with the 16-byte slot XOR-swizzled by (row>>1)&7, which
//    makes the ds_read_b128 fragment reads conflict-free (two 128-B rows share
//    a 256-B bank row; 16-lane service groups then hit 16 distinct slots); the
//    B image orders its rows u-major (row = (n&3)*32 + n/4) so that the
//    column-interleaved tiles {n0+4j+u} -- which make the epilogue a 16-byte
//    store -- still read consecutive rows.  Reads are bounded by buffer
//    descriptors: rows >= m of A and rows >= k of B come back as 0, and
//    garbage * 0 == 0 exactly in integers, so ANY m, n, k works unmasked.
//    Needs lda, ldb multiples of 4 and 4-byte-aligned A, B.  (The first pipelined kernel of the
//    round; kept as an A/B rung, MMH_OPT_IGEMM_MODE 1.)
//  * igemm_s8_simple_kernel: the correctness-first version (single buffer, byte
//    loads at the edges) kept for operands that are not 4-byte aligned, and the
//    independent code path the differential fuzz (tools/fuzz_i8.py) compares against.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ab_build.hpp"
#include "sgemm_tile.hpp"   // block_to_tile, static_for

namespace mmh {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// K-slice geometry of every LDS-DMA int8 kernel here
constexpr int IK = 128;            // k bytes per LDS slice (two 64-deep MFMA steps)
constexpr int ITILE = 128 * IK;    // bytes of one operand image

// --------------------------------------------------------------------------
// K3d / K3t: the tile arithmetic fed entirely by LDS-DMA (K3, the in-kernel-transpose rung it replaced, and K3d's
// packing kernel live in tools/ab/igemm_s8_k3.hpp: tools build only since round 6 -- mode 0 never reached them).
//
// K3 above is instruction-issue-bound: 32 MFMAs of ~17 cycles per slice against
// 12 LDS stores, 32 v_perm_b32 and 8 VGPR loads.  Here B is packed ONCE per call
// into Bt[n][k] (k contiguous, zero-padded to 128 x 128 -- the role packB plays
// inside the reference's CPU MY_MMult, aarch64/MMult_4x4_13.cpp:259-441), and
// both operands then go global -> LDS with `buffer_load_dwordx4 ... lds`: no
// staging registers, no ds_write, no v_perm in the loop.  The DMA destination
// is lane-linear (lane L -> bytes 16 L of a 1 KiB chunk = row L/8, slot L%8 of
// the [row][128 B] image), so the slot swizzle of K3's images is applied on the
// SOURCE side: lane L fetches slot (L%8) ^ ((row>>1)&7) of its row -- same
// 128-byte line, same coalescing.  The B image keeps K3's u-major row order
// by choosing which packed row each lane group fetches.
// --------------------------------------------------------------------------

// N consecutive 1 KiB LDS-DMA chunks (8 image rows each): lane L's 16 bytes land at dst + 1024 j + 16 L.
// (Members of class templates, as Stage's loaders are: hipcc's host pass rejects
// __amdgpu_buffer_rsrc_t in the signature of a function TEMPLATE and in dependent lambdas of a
// __global__ template, and then silently drops the kernel's launch stub.)
template <int N>
struct LdsDma {
  static __device__ __forceinline__ void chunks(__amdgpu_buffer_rsrc_t rsrc, int8_t *dst, const uint32_t (&voff)[N],
                                                int k0) {
#pragma unroll
    for (int j = 0; j < N; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + 8 * j * IK),
                                               16, voff[j], k0, 0, 0);
  }
};
// One template, two configurations <BM, BN, TM> (wave tile = 16 TM rows x 64 columns):
//   <128,128,4>: 4 waves x (64x64), 64 KiB LDS, two workgroups per CU;
//   <256,256,8>: 8 waves x (128x64), 128 KiB LDS, one workgroup per CU.
// int8 MFMAs retire 8x the bytes per cycle of the fp32 ones, so it is operand delivery that
// bounds this kernel: a 128x128 tile needs 64 B/clk/CU from L2 at MFMA peak and 16
// ds_read_b128 per 32 MFMAs per wave; the 256x256 tile halves the former and needs 12.
// Each 128-byte slice is two 64-deep MFMA steps, run as phases of 16 MFMAs whose fragments are read
// during the phase before; the barrier sits before the last phase of a slice (its fragments are in
// registers by then), and right after it the slice after next is requested into the buffer
// just released -- a whole slice ahead of the barrier that needs it.  The slice loop is
// unrolled by two so that the LDS buffer index is a compile-time constant and no address
// arithmetic is left between the MFMAs.
// ABLATE: 0 the kernel; 1..4 timing-only ablations with WRONG results (1 no DMA in the loop,
// 2 no fragment reads in the loop, 3 neither, 4 no C store) -- profiles/r01_ablation.md.
// BTR (K3t): no packed copy of B at all.  `Bt` is then B itself (row-major k x n, `kp` = ldb): its
// 128 x BN byte slice goes into LDS as it lies in memory (LDS-DMA, 16-byte slots XOR-swizzled on
// the source side) and the MFMA fragment -- 16 consecutive k of one column per lane -- is gathered
// by two `ds_read_b64_tr_b8` (gfx950's transposing LDS read; semantics probed with
// tools/probes/ds_read_tr8_probe.hip: within a 16-lane group lane p supplies the address of 8
// bytes, pieces 2j and 2j+1 form row j of an 8 x 16 byte block, lane i receives column i).  The
// MFMA operands are swapped (D = tile^T), which leaves every lane with four consecutive COLUMNS of
// one C row: 16-byte C accesses without the column-interleave trick of the packed path.
template <int BM, int BN, int TM, bool EDGE, int ABLATE, bool BTR = false>
__device__ __forceinline__ void igemm_s8_dma_tile(int m, int n, int k, const int8_t *__restrict__ A, int lda,
                                                  const int8_t *__restrict__ Bt, int kp, int n_pad,
                                                  int32_t *__restrict__ C, int ldc, int accumulate, int nbm,
                                                  int nbn, const float *__restrict__ deq) {
  // deq != nullptr: C receives fp32 = (float)acc * (1 / (deq[0] * deq[1])) -- the dequantisation of a
  // symmetric-quantised GEMM (quant_s8.hpp) fused into the epilogue instead of an int32 round trip
  constexpr int ABL = ABLATE;
  constexpr bool DMA_ON = ABL != 1 && ABL != 3, READS_ON = ABL != 2 && ABL != 3;
  constexpr int TN = 4;                                           // 16-column MFMA tiles per wave
  constexpr int WAVES_N = BN / 64, WAVES = BM / (16 * TM) * WAVES_N;
  constexpr int A_IMG = BM * IK, B_IMG = BN * IK, STAGE = A_IMG + B_IMG;
  constexpr int CA = BM / 8 / WAVES, CB = BN / 8 / WAVES;   // 1 KiB DMA chunks per wave per image
  constexpr int ND = CA + CB;                                     // LDS-DMA instructions per wave and slice
  static_assert(BM % (16 * TM) == 0 && BN % 64 == 0, "wave tile");
  static_assert((BM / 8) % WAVES == 0 && (BN / 8) % WAVES == 0, "DMA chunks per wave");
  extern __shared__ __attribute__((aligned(16))) int8_t ilds[];   // 2 x (A image | B image)

  int tm, tn;
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;

  const int rows_valid = EDGE ? min(BM, m - row0) : BM;
  const bool whole_c = !EDGE || (rows_valid == BM && col0 + BN <= n);
  // packed path: lane (li, g) holds C[crow + 16 t + r][ccol + u] (column-interleaved tiles);
  // BTR: C[crow + 16 t][ccol + 16 u + r] (four consecutive columns, the MFMA's r index)
  const int crow = row0 + wm * 16 * TM + (BTR ? li : 4 * g);
  const int ccol = col0 + wn * 64 + (BTR ? 4 * g : 4 * li);
  typedef int c_vec_u __attribute__((ext_vector_type(4), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, i32x4>;

  i32x4 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // BTR: r plays the tile index u here (one 16-byte vector per tile)
      const int row = crow + 16 * t + (BTR ? 0 : r), col = ccol + (BTR ? 16 * r : 0);
      i32x4 v = {0, 0, 0, 0};
      if (accumulate) {
        if (whole_c) {
          v = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + col);
        } else if (row < m) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (col + u < n) v[u] = C[(size_t)row * ldc + col + u];
        }
      }
      if constexpr (BTR) {
        acc[t][r] = v;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u][r] = v[u];
      }
    }

  const int nk = 2 * ((k + 2 * IK - 1) / (2 * IK));   // slices, rounded up to even (k > 0)
  // descriptors: A bounded at the block's last valid row (rows >= m arrive as zeros;
  // bytes past k are multiplied by Bt's zero padding); Bt is zero-padded to n_pad rows,
  // rows past that (a 256-wide tile over a 128-padded Bt) are cut off by the extent
  const uint32_t ext_a = (uint32_t)((rows_valid - 1) * lda + ((k + 3) & ~3));
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(A + (size_t)row0 * lda), 0, ext_a, 0x00020000);
  // BTR: B (row-major, ldb = kp) from column col0: rows >= k lie past the extent and arrive as
  // zeros; columns >= n read the neighbouring bytes and only feed C columns that are never stored
  const uint32_t ext_b = BTR ? (uint32_t)((k - 1) * kp + ((min(BN, n - col0) + 3) & ~3))
                             : (uint32_t)(min(BN, n_pad - col0) * kp);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(BTR ? Bt + col0 : Bt + (size_t)col0 * kp), 0, ext_b, 0x00020000);
  // wave w moves chunks CA w .. CA w + CA-1 of the A image and likewise of the B image
  const int dr = lane >> 3, dsl = lane & 7;
  uint32_t voff_a[CA], voff_b[CB];
#pragma unroll
  for (int j = 0; j < CA; ++j) {
    const int prow = 8 * (CA * wave + j) + dr;                      // image row
    voff_a[j] = (uint32_t)(prow * lda + 16 * (dsl ^ ((prow >> 1) & 7)));
  }
  // BTR image: [k row][BN bytes]; the 16-byte slot of row r is XORed with btr_swz(r), chosen so that
  // the 16 + 16 pieces of the two lane groups a transposing read serves together (8 rows each,
  // 16 rows apart) fall on 16 different slots of the 256-byte bank row
  auto btr_swz = [](int r) {
    return BN == 256 ? ((r & 7) | (((r >> 4) & 1) << 3)) : (((r >> 1) & 3) | (((r >> 4) & 1) << 2));
  };
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    if constexpr (BTR) {
      constexpr int LPR = BN / 16, RPC = 1024 / BN;                 // lanes per row, rows per 1 KiB chunk
      const int r = RPC * (CB * wave + j) + lane / LPR;             // k row inside the slice
      voff_b[j] = (uint32_t)(r * kp + 16 * ((lane % LPR) ^ btr_swz(r)));
    } else {
      const int prow = 8 * (CB * wave + j) + dr;
      const int nloc = 4 * (prow % (BN / 4)) + prow / (BN / 4);     // u-major image row -> column
      voff_b[j] = (uint32_t)(nloc * kp + 16 * (dsl ^ ((prow >> 1) & 7)));
    }
  }
  // past the last slice the same instructions run against zero-length descriptors: every lane is
  // out of range, zeros land in LDS, nothing is fetched
  const __amdgpu_buffer_rsrc_t null_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(A), 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(Bt), 0, 0, 0x00020000);
  auto dma = [&](int8_t *buf, int kt) {
    const bool live = kt < nk;
    LdsDma<CA>::chunks(live ? rsrc_a : null_a, buf + 8 * CA * wave * IK, voff_a, kt * IK);
    LdsDma<CB>::chunks(live ? rsrc_b : null_b, buf + A_IMG + 8 * CB * wave * IK, voff_b,
                       BTR ? kt * IK * kp : kt * IK);
  };
  // fragment addresses: everything that depends on the lane, per (buffer, MFMA step); tile
  // indices add compile-time constants that fit the ds_read offset field
  const int swz = (li >> 1) & 7;
  uint32_t a_off[2][2], b_off[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      a_off[c][st] = (uint32_t)(c * STAGE + (wm * 16 * TM + li) * IK + 16 * ((4 * st + g) ^ swz));
      b_off[c][st] = (uint32_t)(c * STAGE + A_IMG + (wn * 16 + li) * IK + 16 * ((4 * st + g) ^ swz));
    }
  // BTR: per (buffer, tile u) the address of this lane's piece of rows 16 g + (li >> 1) (+ 64 st + 8 h
  // as an immediate): piece p = li of its group is row p >> 1, half p & 1 of the 8 x 16 block
  uint32_t bt_off[2][TN];
  if constexpr (BTR) {
    const int r = 16 * g + (li >> 1);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int u = 0; u < TN; ++u)
        bt_off[c][u] = (uint32_t)(c * STAGE + A_IMG + r * BN + 16 * ((4 * wn + u) ^ btr_swz(r)) + 8 * (li & 1));
  }
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  // The K loop runs in PHASES of 16 MFMAs: 4 A tiles (one "half" of a 128-row wave tile, or all of a
  // 64-row one) x 4 B tiles of one 64-deep step.  Two register sets each: A fragments alternate from
  // phase to phase, B fragments from step to step.
  constexpr int HALVES = TM / 4, PHASES = 2 * HALVES, NB = (BTR ? 2 : 1) * TN;
  static_assert(TM % 4 == 0, "phases are four A tiles high");
  i32x4 fa[2][4], fb[2][TN];
  auto read_a = [&](auto set_c, auto c_c, auto st_c, auto h_c) {   // A tiles 4h .. 4h+3 of step st, buffer c
    constexpr int SET = decltype(set_c)::value, CB_ = decltype(c_c)::value, ST = decltype(st_c)::value;
    constexpr int H = decltype(h_c)::value;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      fa[SET][t] = *reinterpret_cast<const i32x4 *>(ilds + a_off[CB_][ST] + 16 * (4 * H + t) * IK);
  };
  auto read_b = [&](auto set_c, auto c_c, auto st_c) {             // the B tiles of step st, buffer c
    constexpr int SET = decltype(set_c)::value, CB_ = decltype(c_c)::value, ST = decltype(st_c)::value;
#pragma unroll
    for (int ju = 0; ju < TN; ++ju) {
      if constexpr (BTR) {
        typedef __attribute__((address_space(3))) i32x2 *lds_v2;
        const i32x2 lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2)(ilds + bt_off[CB_][ju] + 64 * ST * BN));
        const i32x2 hi =
            __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2)(ilds + bt_off[CB_][ju] + (64 * ST + 8) * BN));
        fb[SET][ju] = i32x4{lo[0], lo[1], hi[0], hi[1]};
      } else {
        fb[SET][ju] = *reinterpret_cast<const i32x4 *>(ilds + b_off[CB_][ST] + ju * (BN / 4) * IK);
      }
    }
  };
  auto mfma_phase = [&](auto aset_c, auto bset_c, auto h_c) {
    constexpr int AS = decltype(aset_c)::value, BS = decltype(bset_c)::value, H = decltype(h_c)::value;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ju = 0; ju < TN; ++ju)
        acc[4 * H + t][ju] =
            BTR ? __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[BS][ju], fa[AS][t], acc[4 * H + t][ju], 0, 0, 0)
                : __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[AS][t], fb[BS][ju], acc[4 * H + t][ju], 0, 0, 0);
  };
  constexpr std::integral_constant<int, 0> i0{};
  constexpr std::integral_constant<int, 1> i1{};

  dma(ilds, 0);
  dma(ilds + STAGE, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  read_a(i0, i0, i0, i0);
  read_b(i0, i0, i0);

  // One slice.  Every slice runs the same code (no peeled tail: the compiler spills the
  // accumulators around peeled copies): the slice count is rounded up to even -- Bt is
  // zero-padded to a multiple of 256 k's (B's descriptor returns zeros past row k) and whatever A's
  // descriptor returns there is multiplied by those zeros -- and past the end the DMAs run against
  // zero-length descriptors, which costs the instructions but no memory traffic.
  //   phases 0 .. PHASES-2: MFMAs, reading the next phase's fragments from this buffer underneath;
  //   barrier             : every wave has finished reading this buffer, slice kt + 1 has landed
  //                         in the other one;
  //   last phase          : MFMAs; underneath, slice kt + 2 is requested into THIS buffer -- a whole
  //                         slice ahead of the barrier that needs it -- and slice kt + 1's first
  //                         fragments are read from the other one.
  auto slice = [&](int kt, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr std::integral_constant<int, CUR> cur{};
    constexpr std::integral_constant<int, CUR ^ 1> oth{};
    static_for<PHASES>([&](auto pc) {
      constexpr int P = decltype(pc)::value, S = P / HALVES, H = P % HALVES;
      constexpr std::integral_constant<int, P & 1> aset{};
      constexpr std::integral_constant<int, (P + 1) & 1> anext{};
      constexpr std::integral_constant<int, S & 1> bset{};
      constexpr std::integral_constant<int, H> half{};
      if constexpr (P + 1 < PHASES) {
        constexpr int NS = (P + 1) / HALVES, NH = (P + 1) % HALVES;      // the next phase
        constexpr std::integral_constant<int, NS> nstep{};
        if (READS_ON) {
          read_a(anext, cur, nstep, std::integral_constant<int, NH>{});
          if constexpr (NH == 0) read_b(std::integral_constant<int, NS & 1>{}, cur, nstep);
        }
        mfma_phase(aset, bset, half);
        constexpr int nr = READS_ON ? 4 + (NH == 0 ? NB : 0) : 0;
        static_for<16>([&](auto ic) {   // pipeline description: the reads dealt out between the MFMAs
          constexpr int I = decltype(ic)::value, R = (I + 1) * nr / 16 - I * nr / 16;
          if constexpr (R > 0) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);   // DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
        });
      } else {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // bare: __syncthreads() would add nothing but a second wait
        __builtin_amdgcn_sched_barrier(0);
        if (DMA_ON) dma(ilds + CUR * STAGE, kt + 2);
        if (READS_ON) {
          read_b(i0, oth, i0);
          read_a(i0, oth, i0, i0);
        }
        mfma_phase(aset, bset, half);
        constexpr int nr = READS_ON ? 4 + NB : 0, nd = DMA_ON ? ND : 0;
        static_for<16>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          constexpr int D = (I + 1) * nd / 16 - I * nd / 16, R = (I + 1) * nr / 16 - I * nr / 16;
          if constexpr (D > 0) __builtin_amdgcn_sched_group_barrier(0x020, D, 0);   // LDS-DMA
          if constexpr (R > 0) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);   // DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
        });
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  for (int kt = 0; kt < nk; kt += 2) {
    slice(kt, i0);
    slice(kt + 1, i1);
  }

  const float deq_inv = deq ? 1.0f / (deq[0] * deq[1]) : 0.0f;
  auto to_c = [&](i32x4 v) {
    if (deq) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t f = {(float)v[0] * deq_inv, (float)v[1] * deq_inv, (float)v[2] * deq_inv, (float)v[3] * deq_inv};
      v = __builtin_bit_cast(i32x4, f);
    }
    return v;
  };
  auto put = [&](int row, int col, i32x4 v) {
    if (whole_c) {
      *reinterpret_cast<c_vec *>(C + (size_t)row * ldc + col) = v;
    } else if (row < m) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (col + u < n) C[(size_t)row * ldc + col + u] = v[u];
    }
  };
  if constexpr (BTR && ABL == 0) {
    // In place, a lane holds tile[16 t + li][16 u + 4 g .. + 3]: stored as they are, the sixteen lanes of a quarter-wave
    // hit sixteen ROWS with 16 bytes each (64 requests per store; round 6's timeline of K3p: 7.5 - 14 us to issue a
    // 256x256 tile's stores).  Out through 4 KiB of the -- now idle -- ring per wave instead (igemm_s8_pp.hpp, the same
    // transposer): lane l stores tile[16 t + 4 q + l / 16][4 (l % 16) .. + 3], a quarter-wave 256 contiguous bytes.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no LDS-DMA may land in what is reused below)
    __syncthreads();                                    // every wave is past its last fragment read
    int8_t *cx = ilds + wave * 4096;
    const int eg = lane >> 4, eli = lane & 15;
    const int drow = row0 + wm * 16 * TM + eg, dcol = col0 + wn * 64 + 4 * eli;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
      for (int u = 0; u < TN; ++u) *reinterpret_cast<i32x4 *>(cx + li * 256 + 16 * ((4 * u + g) ^ li)) = to_c(acc[t][u]);
      i32x4 out[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = 4 * q + eg;
        out[q] = *reinterpret_cast<const i32x4 *>(cx + rr * 256 + 16 * (eli ^ rr));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) put(drow + 16 * t + 4 * q, dcol, out[q]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = crow + 16 * t + (BTR ? 0 : r), col = ccol + (BTR ? 16 * r : 0);
        i32x4 v = BTR ? acc[t][r] : i32x4{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
        if (ABL == 4 && (v[0] ^ v[1] ^ v[2] ^ v[3]) != 0x7ffffff1) continue;
        put(row, col, to_c(v));
      }
  }
}

// The kernel proper is a thin shell around the __device__ template (see LdsDma above).
template <int BM, int BN, int TM, bool EDGE, int ABLATE, bool BTR = false>
__global__ void __launch_bounds__(BM / (16 * TM) * (BN / 64) * 64, (BM == 128 && BN == 128) ? 2 : 1)
igemm_s8_dma_kernel(int m, int n, int k, const int8_t *__restrict__ A, int lda,
                    const int8_t *__restrict__ Bt, int kp, int n_pad, int32_t *__restrict__ C, int ldc,
                    int accumulate, int nbm, int nbn, const float *__restrict__ deq) {
  igemm_s8_dma_tile<BM, BN, TM, EDGE, ABLATE, BTR>(m, n, k, A, lda, Bt, kp, n_pad, C, ldc, accumulate, nbm, nbn,
                                                   deq);
}

// --------------------------------------------------------------------------
// The correctness-first kernel (any alignment).
// --------------------------------------------------------------------------
constexpr int IBK = 64;      // k bytes per LDS slice
constexpr int IPITCH = 80;   // LDS row pitch in bytes (64 + 16 pad, 16-B aligned)

template <bool EDGE>
__global__ void __launch_bounds__(256)
igemm_s8_simple_kernel(int m, int n, int k, const int8_t *__restrict__ A, int lda,
                       const int8_t *__restrict__ B, int ldb, int32_t *__restrict__ C, int ldc,
                       int accumulate, int nbm, int nbn) {
  constexpr int BM = 128, BN = 128;
  __shared__ __attribute__((aligned(16))) int8_t lds[(BM + BN) * IPITCH];
  int8_t *As = lds, *Bt = lds + BM * IPITCH;

  const int tile = blockIdx.x;
  const int tm = tile / nbn, tn = tile % nbn;
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, g = lane >> 4;

  const int crow = row0 + wm * 64 + 4 * g;     // + 16 t + r
  const int ccol = col0 + wn * 64 + 4 * li;    // .. +3 (u)

  i32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 16 * t + r;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int v = 0;
        if (accumulate && (!EDGE || (row < m && ccol + u < n))) v = C[(size_t)row * ldc + ccol + u];
        acc[t][u][r] = v;
      }
    }

  const int nk = (k + IBK - 1) / IBK;
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = kt * IBK;
    // ---- stage A: 128 rows x 64 bytes, two 16-byte vectors per thread ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = (tid >> 2) + 64 * pass, ch = tid & 3;
      i32x4 v = {0, 0, 0, 0};
      if (!EDGE) {
        v = *reinterpret_cast<const i32x4 *>(A + (size_t)(row0 + r) * lda + k0 + 16 * ch);
      } else if (row0 + r < m) {
        int8_t tmp[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          const int kk = k0 + 16 * ch + b;
          tmp[b] = kk < k ? A[(size_t)(row0 + r) * lda + kk] : (int8_t)0;
        }
        v = *reinterpret_cast<i32x4 *>(tmp);
      }
      *reinterpret_cast<i32x4 *>(As + r * IPITCH + 16 * ch) = v;
    }
    // ---- stage B: 64 k x 128 n bytes, 4x4 byte blocks transposed in registers ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int kb = (tid >> 5) + 8 * pass, nb = tid & 31;
      uint32_t rr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k0 + 4 * kb + j;
        if (!EDGE) {
          rr[j] = *reinterpret_cast<const uint32_t *>(B + (size_t)kk * ldb + col0 + 4 * nb);
        } else {
          uint32_t w = 0;
          if (kk < k) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int col = col0 + 4 * nb + u;
              if (col < n) w |= (uint32_t)(uint8_t)B[(size_t)kk * ldb + col] << (8 * u);
            }
          }
          rr[j] = w;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t w = ((rr[0] >> (8 * u)) & 0xffu) | (((rr[1] >> (8 * u)) & 0xffu) << 8) |
                           (((rr[2] >> (8 * u)) & 0xffu) << 16) | (((rr[3] >> (8 * u)) & 0xffu) << 24);
        *reinterpret_cast<uint32_t *>(Bt + (4 * nb + u) * IPITCH + 4 * kb) = w;
      }
    }
    __syncthreads();
    // ---- one 64-deep MFMA step: 4 A reads + 4 B reads feed 16 MFMAs ----
    i32x4 a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      a[t] = *reinterpret_cast<const i32x4 *>(As + (wm * 64 + 16 * t + li) * IPITCH + 16 * g);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      b[u] = *reinterpret_cast<const i32x4 *>(Bt + (wn * 64 + 4 * li + u) * IPITCH + 16 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t], b[u], acc[t][u], 0, 0, 0);
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 16 * t + r;
      i32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
      if (!EDGE) {
        *reinterpret_cast<i32x4 *>(C + (size_t)row * ldc + ccol) = v;
      } else if (row < m) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ccol + u < n) C[(size_t)row * ldc + ccol + u] = v[u];
      }
    }
}

template <int BM, int BN, int TM, bool EDGE, int ABLATE, bool BTR = false>
inline hipError_t launch_igemm_s8_dma_edge(int m, int n, int k, const int8_t *A, int lda, const int8_t *Bt,
                                           int kp, int n_pad, int32_t *C, int ldc, int acc, hipStream_t s,
                                           const float *deq = nullptr) {
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  constexpr int threads = BM / (16 * TM) * (BN / 64) * 64;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * IK;
  if (lds > 64 * 1024) {   // > 64 KiB of dynamic LDS must be opted into (remembered per device)
    const hipError_t e = opt_in_big_lds(reinterpret_cast<const void *>(&igemm_s8_dma_kernel<BM, BN, TM, EDGE, ABLATE, BTR>),
                                lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((igemm_s8_dma_kernel<BM, BN, TM, EDGE, ABLATE, BTR>), dim3((unsigned)(nbm * nbn)), dim3(threads),
                     lds, s, m, n, k, A, lda, Bt, kp, n_pad, C, ldc, acc, nbm, nbn, deq);
  return hipGetLastError();
}

template <int BM, int BN, int TM, bool BTR = false>
inline hipError_t launch_igemm_s8_dma(int m, int n, int k, const int8_t *A, int lda, const int8_t *Bt, int kp,
                                      int n_pad, int32_t *C, int ldc, int acc, hipStream_t s,
                                      const float *deq = nullptr) {
  const bool c_fast = (m % BM == 0) && (n % BN == 0) && (ldc % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  return c_fast
             ? launch_igemm_s8_dma_edge<BM, BN, TM, false, 0, BTR>(m, n, k, A, lda, Bt, kp, n_pad, C, ldc, acc, s,
                                                                  deq)
             : launch_igemm_s8_dma_edge<BM, BN, TM, true, 0, BTR>(m, n, k, A, lda, Bt, kp, n_pad, C, ldc, acc, s,
                                                                 deq);
}

// K3t (B read in place) needs 4-byte aligned operands and byte offsets inside the descriptors' 2 GiB
inline bool igemm_s8_inplace_ok(const int8_t *A, int lda, const int8_t *B, int ldb, int k) {
  const size_t lim = (1ull << 31) - 4096;
  return (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 3) == 0) && (ldb % 4 == 0) &&
         ((reinterpret_cast<uintptr_t>(B) & 3) == 0) && ((size_t)256 * lda + k) < lim &&
         ((size_t)(k + 256) * ldb + 256) < lim;
}
// C_f32 = (float)(A x B) * (1 / (scales[0] * scales[1])) with the dequantisation in the epilogue; in-place
// kernel only (the caller checks igemm_s8_inplace_ok and otherwise runs the two-pass form).
// Tile choice of the in-place kernels: 256x256 tiles (K3p) run one per CU, 128x128 tiles (K3t) two per CU, and a full
// round of 128x128 tiles takes 0.58 of a 256x256 round's time for half its work (4096^3: 68.5 us in two rounds against
// 58.7 in one).  Rounds 3 - 5 never chose the big tile below one tile per CU; with the C stores coalesced (round 6) a
// part-filled chip of big tiles beats two rounds of small ones: 3072^3 31.8 against 43.1 us, 3584^3 42.9 / 53.0, 6144^3
// 205 / 217 -- and a single round of small tiles wins where there is one: 2048^3 14.1 / 22.5, 1024^3 9.8 / 13.1
// (profiles/r06_i8_persist_ab.txt).
inline bool igemm_s8_big_tile(int m, int n, int num_cus) {
  const long tiles256 = (long)((m + 255) / 256) * ((n + 255) / 256);
  const long tiles128 = (long)((m + 127) / 128) * ((n + 127) / 128);
  const long r256 = (tiles256 + num_cus - 1) / num_cus, r128 = (tiles128 + 2L * num_cus - 1) / (2L * num_cus);
  return (double)r256 <= 0.6 * (double)r128 + 1e-9;
}

inline hipError_t launch_igemm_s8_dequant(int m, int n, int k, const int8_t *A, int lda, const int8_t *B, int ldb,
                                          float *C, int ldc, const float *scales, hipStream_t s, int num_cus) {
  int32_t *Ci = reinterpret_cast<int32_t *>(C);
  if (igemm_s8_big_tile(m, n, num_cus))
    return launch_igemm_s8_dma<256, 256, 8, true>(m, n, k, A, lda, B, ldb, n, Ci, ldc, 0, s, scales);
  return launch_igemm_s8_dma<128, 128, 4, true>(m, n, k, A, lda, B, ldb, n, Ci, ldc, 0, s, scales);
}

// mode: 0 = K3t (B read in place by transposing LDS reads; the tile by igemm_s8_big_tile) when the operands are 4-byte
//           aligned and inside the descriptors' window, else the simple kernel; 5 / 6 = K3t with 128x128 / 256x256 tiles
//       forced (A/B switches); 2 (and anything K3t cannot take) = the simple kernel.
// (The ping-pong kernel K3p is launched by igemm.hip; K3 and the packed-B kernel K3d -- modes 1, 3, 4, and the timing-only
// ablations 10..13 -- by tools/ab/igemm_s8_k3.hpp in the tools build.)
inline hipError_t launch_igemm_s8(int m, int n, int k, const int8_t *A, int lda, const int8_t *B,
                                  int ldb, int32_t *C, int ldc, int acc, hipStream_t s, int mode = 0, int num_cus = 256) {
  const int nbm = (m + 127) / 128, nbn = (n + 127) / 128;
  dim3 grid((unsigned)(nbm * nbn)), block(256);
  const bool shape_ok = (m % 128 == 0) && (n % 128 == 0);
  if ((mode == 0 || mode == 5 || mode == 6) && igemm_s8_inplace_ok(A, lda, B, ldb, k)) {
    // K3t: B read in place (no packing, no workspace)
    const bool big = mode == 6 || (mode == 0 && igemm_s8_big_tile(m, n, num_cus));
    if (big) return launch_igemm_s8_dma<256, 256, 8, true>(m, n, k, A, lda, B, ldb, n, C, ldc, acc, s);
    return launch_igemm_s8_dma<128, 128, 4, true>(m, n, k, A, lda, B, ldb, n, C, ldc, acc, s);
  }
  const bool fast = shape_ok && (k % IBK == 0) && (lda % 16 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(B) & 3) == 0) &&
                    ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  if (fast)
    hipLaunchKernelGGL(igemm_s8_simple_kernel<false>, grid, block, 0, s, m, n, k, A, lda, B, ldb, C,
                       ldc, acc, nbm, nbn);
  else
    hipLaunchKernelGGL(igemm_s8_simple_kernel<true>, grid, block, 0, s, m, n, k, A, lda, B, ldb, C,
                       ldc, acc, nbm, nbn);
  return hipGetLastError();
}

}  // namespace mmh
