// igemm_s8_pp.hpp -- K3p: the int8 GEMM's 256x256 tile with its two wave groups in PING-PONG, PERSISTENT over tiles.
//
// Same arithmetic, same LDS images and the same fragment reads as K3t (igemm_s8.hpp: B read in place by
// LDS-DMA, fragments by ds_read_b64_tr_b8, v_mfma_i32_16x16x64_i8, exact integers) -- what changes is WHEN
// each wave does what.  In K3t all eight waves run the same phase at once: each SIMD's two waves both want
// the matrix pipe, both interleave their fragment reads with their MFMAs, a whole slice of LDS-DMA (64 pieces
// per workgroup) is requested in one phase and drained with `vmcnt(0)` once per slice; the matrix pipe is busy
// 45-61 % of the launch (profiles/r02_igemm_s8_rocprofv3.json).  Here (cdna guide, "8-phase" schedule):
//   * the workgroup's two wave groups -- waves 0-3 (C rows 0-127) and waves 4-7 (rows 128-255), one wave of
//     each per SIMD -- run ONE BARRIER apart.  A phase of a wave is  [R: fragment reads for 32 MFMAs, four
//     LDS-DMA pieces] barrier [M: the 32 MFMAs, s_setprio 1] barrier , so while one wave of a SIMD sits in M
//     the other is in R: the pipe always has a wave that does nothing but MFMAs, the LDS a wave that does
//     nothing but reads;
//   * the LDS-DMA of a slice is dealt out over the phases, four pieces per wave and phase, each region of the
//     double buffer re-requested as soon as its last reader is done, two or three phases ahead of its first
//     reader; the wait is a counted `vmcnt(4)` per phase -- never 0 in the loop -- one phase and one barrier
//     ahead of the reads it guards.
// Regions of a slice's image: A0 / A1 = rows 0-127 / 128-255 of the A image (read by group 0 / group 1 only),
// B0 / B1 = k rows 0-63 / 64-127 of the B image (MFMA step 0 / 1).
//
// Round 6: PERSISTENT.  Rounds 3-5 launched one workgroup per tile; a shape with several tiles per CU (8192^3:
// four) paid ~25 us per round of tiles beside an 86 us loop: the 64 MB of C (13 us at the 4.9 TB/s the chip
// writes at), the workgroup's retirement (its 128 KiB of LDS must be free before the next one is placed), the
// launch of the next and its prologue, one after the other.  Now min(tiles, CUs) workgroups walk the tiles
// (tile = blockIdx.x + j * gridDim.x: the XCD / raster map of block_to_tile applied to the virtual index, so an
// XCD's CUs keep working on one band of tiles); at a tile's end the two groups fall back into step, the NEXT
// tile's prologue (ten LDS-DMA pieces per wave) is requested FIRST, then the 32 C stores of the finished tile
// leave, and the next tile's first two phases wait with `vmcnt(4 + 32)` -- returns are in order, so "all but the
// youngest 36" still means "every piece requested in front of the stores has landed" without waiting for one
// store.  (A store's data registers may be overwritten as soon as it has issued; only the third phase's wait,
// which guards pieces requested BEHIND the stores, has to see them acknowledged.)  The register file holds one
// 256x256 int32 tile per CU and not two (256 KiB of the CU's 512), so the store of a CU's LAST tile overlaps
// nothing: at 4096^3 -- one tile per CU -- this kernel is what round 5's was.
//
// MFMA_K = 32 builds the same kernel on `v_mfma_i32_16x16x32_i8`, the instruction BASELINE.json configs[4]
// names: each 64-deep step is two 32-deep instructions on the low / high 8 bytes of the lanes' 16 (both operands
// use the same lane -> k map, so the sum runs over the same 64 products) -- bit-identical, half the rate of the
// pipe (profiles/r04_i8_instr_ab.md: 2.33 against 3.91 POPS on random operands).  MMH_OPT_IGEMM_MODE 7.
// Measured: round 3 (profiles/r03_igemm_s8_ksweep.txt; M = N = 4096, K swept) 17.1 us + a loop at 3.18 POPS = 0.83 of what
// the matrix pipe sustains on random operands at the power-managed clock; round 6 (profiles/r06_notes.md section 1): the
// "fixed" part was the C stores' issue time -- with the transposer 4096^3 runs 56.5 us (2.43 POPS), 8192^3 2.65 - 2.74 POPS.
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

// One 64-deep MFMA step of a 16x16 tile: the double-rate instruction, or the config-named 32-deep one twice.
template <int MFMA_K>
__device__ __forceinline__ i32x4 pp_mfma_step(i32x4 b, i32x4 a, i32x4 c) {
  if constexpr (MFMA_K == 64) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c, 0, 0, 0);
  } else {
    typedef long pp_i64x2 __attribute__((ext_vector_type(2)));
    const pp_i64x2 b2 = __builtin_bit_cast(pp_i64x2, b), a2 = __builtin_bit_cast(pp_i64x2, a);
    c = __builtin_amdgcn_mfma_i32_16x16x32_i8(b2[0], a2[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_i32_16x16x32_i8(b2[1], a2[1], c, 0, 0, 0);
  }
}

#ifdef MMH_DMA_TIMELINE
// timeline build only (tools/i8_timeline.py): waves 0 and 4 of every workgroup keep eight wall-clock stamps per tile in
// scalar registers and write them behind the loop's last wait -- never between the C stores and the counted waits
__device__ unsigned long long *g_i8_stamps = nullptr;
#define MMH_I8_STAMP(i) do { if (g_i8_stamps) st[i] = wall_clock64(); } while (0)
#else
#define MMH_I8_STAMP(i) do { } while (0)
#endif

// What a tile of K3p needs besides its accumulators: where it sits, and the two descriptors its slices are requested through
// (A bounded at the block's last valid row, B -- row-major, in place -- at row k: rows past them come back as 0).
struct PpTileAt {
  int row0, col0, rows_valid;
  bool whole_c;
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
};

template <bool EDGE, bool DEQ, int MFMA_K = 64>
__global__ void __launch_bounds__(512, 1)
igemm_s8_pp_kernel(int m, int n, int k, const int8_t *__restrict__ A, int lda, const int8_t *__restrict__ B, int ldb,
                   int32_t *__restrict__ C, int ldc, int accumulate, int nbm, int nbn, const float *__restrict__ deq) {
  constexpr int BM = 256, BN = 256, TM = 8, TN = 4;
  constexpr int A_IMG = BM * IK, B_IMG = BN * IK, STAGE = A_IMG + B_IMG;   // 32 KiB + 32 KiB
  constexpr int STORES = TM * TN;                                           // C stores per wave and tile (whole tiles)
  extern __shared__ __attribute__((aligned(16))) int8_t ilds[];            // 2 x STAGE = 128 KiB, + 8 x 4 KiB (the C transposer)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;     // wm = the wave's group (0: older half, 1: younger half)
  const int li = lane & 15, g = lane >> 4;
  const int ntiles = nbm * nbn;
  typedef int c_vec_u __attribute__((ext_vector_type(4), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, i32x4>;

  const int nk = 2 * ((k + 2 * IK - 1) / (2 * IK));   // slices, rounded up to even (k > 0)
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
  const uint32_t lds_base = (uint32_t)(uintptr_t)ilds;
  constexpr std::integral_constant<int, 0> i0{};
  constexpr std::integral_constant<int, 1> i1{};
  constexpr std::integral_constant<int, 2> i2{};
  constexpr std::integral_constant<int, 3> i3{};
  const float deq_inv = DEQ ? 1.0f / (deq[0] * deq[1]) : 0.0f;

  auto tile_at = [&](int tile) {
    int tm, tn;
    block_to_tile(tile, ntiles, nbm, nbn, tm, tn);
    PpTileAt t;
    t.row0 = tm * BM;
    t.col0 = tn * BN;
    t.rows_valid = EDGE ? min(BM, m - t.row0) : BM;
    t.whole_c = !EDGE || (t.rows_valid == BM && t.col0 + BN <= n);
    const uint32_t ext_a = (uint32_t)((t.rows_valid - 1) * lda + ((k + 3) & ~3));
    const uint32_t ext_b = (uint32_t)((k - 1) * ldb + ((min(BN, n - t.col0) + 3) & ~3));
    t.rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(A + (size_t)t.row0 * lda), 0, ext_a, 0x00020000);
    t.rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(B + t.col0), 0, ext_b, 0x00020000);
    return t;
  };
  // request region `reg` (0: A0, 1: A1, 2: B0, 3: B1) of slice kt of tile `t` into buffer `buf`
  auto request = [&](const PpTileAt &t, auto reg_c, int8_t *buf, int kt) {
    constexpr int REG = decltype(reg_c)::value, HS = REG & 1;
    const bool live = kt < nk;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (REG < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? t.rsrc_a : null_a,
                                                 (__attribute__((address_space(3))) void *)(buf + (128 * HS + 8 * (2 * wave + j)) * IK),
                                                 16, voff_a[HS][j], kt * IK, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? t.rsrc_b : null_b,
                                                 (__attribute__((address_space(3))) void *)(buf + A_IMG + (64 * HS + 4 * (2 * wave + j)) * BN),
                                                 16, voff_b[HS][j], kt * IK * ldb, 0, 0);
    }
  };
  // a tile's prologue: slice 0 whole and B0 of slice 1 -- ten pieces per wave, the six oldest are what phase (0, 0) reads
  auto request_prologue = [&](const PpTileAt &t) {
    request(t, i2, ilds, 0);
    request(t, i0, ilds, 0);
    request(t, i1, ilds, 0);
    request(t, i3, ilds, 0);
    request(t, i2, ilds + STAGE, 1);
  };

  i32x4 acc[TM][TN];
  i32x4 fa[TM], fb[TN];
  pp_i32x2 fb_lo[TN], fb_hi[TN];

  PpTileAt cur = tile_at(blockIdx.x);
  request_prologue(cur);
#ifdef MMH_DMA_TIMELINE
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tile_it = 0;
#endif
  bool stores_in_flight = false;   // this wave's C stores of the previous tile sit in the memory queue BEHIND the prologue
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u) acc[t][u] = i32x4{0, 0, 0, 0};
    // The transposing reads are spelled in inline asm: through the builtin hipcc cannot tell what the read may alias
    // and puts `s_waitcnt vmcnt(0)` in front of it whenever an LDS-DMA is in flight -- which here is always, by design.
    // (cdna guide 5.7, form (iii): "=v" loads, a wait-only statement, sched_barrier(0) before the first consumer --
    // all three sit in `phase2` below, in front of the barrier that precedes the MFMAs.)
    // Request schedule, four pieces per wave and phase, issued at the TOP of the phase's R section:
    //   phase 0 of slice t: A0 and A1 of slice t+1 (other buffer; its A regions were last read in slice t-1)
    //   phase 1 of slice t: B1 of slice t+1 (other buffer), B0 of slice t+2 (this buffer, last read in phase 0)
    // every region two or three phases ahead of its first reader; the wait, after the phase's own reads, is
    // `vmcnt(4)`: everything but the four pieces just requested -- which is what the NEXT phase's reads need.
    // With the previous tile's STORES stores queued between the prologue and this tile's first requests, the prologue
    // wait and phase (0, 0)'s wait leave them out of the count: vmcnt(4 + STORES).
    __builtin_amdgcn_sched_barrier(0);
    MMH_I8_STAMP(0);
    if (stores_in_flight) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    static_assert(4 + STORES == 36, "the counted wait behind a tile's C stores");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // the younger group runs one barrier behind from here on
    MMH_I8_STAMP(1);
    auto phase2 = [&](int kt, auto cur_c, auto s_c, bool behind_stores) {
      constexpr int CUR = decltype(cur_c)::value, S = decltype(s_c)::value;
      int8_t *mine = ilds + CUR * STAGE, *other = ilds + (CUR ^ 1) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (S == 0) {
        request(cur, i0, other, kt + 1);
        request(cur, i1, other, kt + 1);
      } else {
        request(cur, i3, other, kt + 1);
        request(cur, i2, mine, kt + 2);
      }
#pragma unroll
      for (int t = 0; t < TM; ++t) fa[t] = *reinterpret_cast<const i32x4 *>(ilds + a_off[CUR][S] + 16 * t * IK);
#pragma unroll
      for (int u = 0; u < TN; ++u)
        ds_read_tr8_pair<64 * S * BN, (64 * S + 8) * BN>(lds_base + bt_off[CUR][u], fb_lo[u], fb_hi[u]);
      if (behind_stores) asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < TN; ++u) fb[u] = i32x4{fb_lo[u][0], fb_lo[u][1], fb_hi[u][0], fb_hi[u][1]};
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = pp_mfma_step<MFMA_K>(fb[u], fa[t], acc[t][u]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt < nk; kt += 2) {
      phase2(kt, i0, i0, stores_in_flight && kt == 0);   // (the tile's first phase: the only wait that may have to look past the stores)
#ifdef MMH_DMA_TIMELINE
      if (kt == 0) MMH_I8_STAMP(2);
#endif
      phase2(kt, i0, i1, false);
#ifdef MMH_DMA_TIMELINE
      if (kt == 0) MMH_I8_STAMP(3);
#endif
      phase2(kt + 1, i1, i0, false);
#ifdef MMH_DMA_TIMELINE
      if (kt == 0) MMH_I8_STAMP(4);
#endif
      phase2(kt + 1, i1, i1, false);
#ifdef MMH_DMA_TIMELINE
      if (kt == 0) MMH_I8_STAMP(5);
#endif
    }
    MMH_I8_STAMP(6);
    if (wm == 0) __builtin_amdgcn_s_barrier();   // the older group's matching barrier: both groups in step, every fragment read done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-length tail requests (and, by now, every older store)
#ifdef MMH_DMA_TIMELINE
    MMH_I8_STAMP(7);
    if (g_i8_stamps && lane == 0 && (wave & 3) == 0 && tile_it < 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) g_i8_stamps[(((size_t)blockIdx.x * 2 + wm) * 16 + tile_it) * 8 + i] = st[i];
    }
    ++tile_it;
#endif

    // the next tile's prologue goes out in front of this tile's C
    const int next = tile + (int)gridDim.x;
    const PpTileAt done = cur;
    if (next < ntiles) {
      cur = tile_at(next);
      request_prologue(cur);
    }
    __builtin_amdgcn_sched_barrier(0);
    // The accumulators leave through a per-wave 4 KiB transposer behind the ring.  Lane (li, g) holds
    // tile[16 t + li][16 u + 4 g .. + 3] (the MFMA operands are swapped: D = tile^T): stored as they are, the sixteen
    // lanes of a quarter-wave hit sixteen ROWS with 16 bytes each -- 64 requests per instruction, and a tile's 256
    // stores took 7.5 - 14 us to ISSUE (tools/i8_timeline.py: all of the "fixed" 17 us per tile that rounds 3 - 5 put down
    // to HBM).  Through the transposer lane l stores tile[16 t + 4 q + l / 16][4 (l % 16) .. + 3]: a quarter-wave writes
    // 256 contiguous bytes of one row, an instruction four rows of two whole cache lines each.  (16-byte slot of row rr
    // XORed with rr: the quarter-waves of both the writes and the reads hit sixteen different slots.)
    // (the lane index behind an empty asm: what is derived from it below -- the transposer's eight addresses, C's -- must
    // be recomputed per tile, not hoisted over the K loop, whose 234 registers have no room for them)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int eli = ln & 15, eg = ln >> 4;
    const int drow = done.row0 + wm * 128 + eg, dcol = done.col0 + wn * 64 + 4 * eli;
    int8_t *cx = ilds + 2 * STAGE + wave * 4096;
    bool counted = done.whole_c;   // whole tiles issue exactly STORES stores per wave and nothing else
    auto put = [&](int row, int col, i32x4 v) {
      if (done.whole_c) {
        // (non-temporal stores were measured here: the fixed part of a 4096^3 launch grows from 17.1 to 21.2 us)
        *reinterpret_cast<c_vec *>(C + (size_t)row * ldc + col) = v;
      } else if (row < m) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < n) C[(size_t)row * ldc + col + e] = v[e];
      }
    };
    auto fetch = [&](int row, int col) {
      i32x4 o = i32x4{0, 0, 0, 0};
      if (done.whole_c) {
        o = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + col);
      } else if (row < m) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < n) o[e] = C[(size_t)row * ldc + col + e];
      }
      return o;
    };
    const bool add_old = !DEQ && accumulate;
    // C += : the old values are fetched HERE and added on the way out -- not loaded into the accumulators in front of the
    // loop: hipcc then guards every later use of them with a wait of its own count (`vmcnt(31)` in front of each store,
    // `vmcnt(35..4)` inside the first phase's MFMAs: the stores would be waited for after all), and a sum INTO the
    // accumulators here costs a second copy of them (spills inside the loop).  Integer sums: the order changes nothing.
    if (add_old) counted = false;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        i32x4 v = acc[t][u];
        if constexpr (DEQ) {
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          const f32x4_t f = {(float)v[0] * deq_inv, (float)v[1] * deq_inv, (float)v[2] * deq_inv, (float)v[3] * deq_inv};
          v = __builtin_bit_cast(i32x4, f);
        }
        *reinterpret_cast<i32x4 *>(cx + eli * 256 + 16 * ((4 * u + eg) ^ eli)) = v;
      }
      i32x4 out[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = 4 * q + eg;
        out[q] = *reinterpret_cast<const i32x4 *>(cx + rr * 256 + 16 * (eli ^ rr));
      }
      if (add_old) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] += fetch(drow + 16 * t + 4 * q, dcol);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) put(drow + 16 * t + 4 * q, dcol, out[q]);
      __builtin_amdgcn_sched_barrier(0);   // (one row of tiles through the transposer at a time)
    }
    // a ragged tile's store count varies, an accumulating one has loads in the queue: their successor waits them out
    stores_in_flight = counted;
    if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}

// `grid_cap`: workgroups at most (the CUs of the device: one persistent workgroup each); 0 = one workgroup per tile,
// round 5's launch form (MMH_OPT_IGEMM_MODE 9, the A/B switch).
template <int MFMA_K = 64>
inline hipError_t launch_igemm_s8_pp(int m, int n, int k, const int8_t *A, int lda, const int8_t *B, int ldb, int32_t *C,
                                     int ldc, int acc, hipStream_t s, int grid_cap, const float *deq = nullptr) {
  constexpr int BM = 256, BN = 256;
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * IK + 8 * 4096;   // the ring and the eight waves' C transposers: 160 KiB
  const bool c_fast = (m % BM == 0) && (n % BN == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  // the XCD map of block_to_tile reads the XCD off the (virtual) block index: a persistent grid is a multiple of NXCD
  int grid = nbm * nbn;
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap >= NXCD ? grid_cap / NXCD * NXCD : grid_cap;
#define MMH_PP_LAUNCH(E, D)                                                                                           \
  do {                                                                                                                \
    const hipError_t e = opt_in_big_lds(reinterpret_cast<const void *>(&igemm_s8_pp_kernel<E, D, MFMA_K>), lds);      \
    if (e != hipSuccess) return e;                                                                                    \
    hipLaunchKernelGGL((igemm_s8_pp_kernel<E, D, MFMA_K>), dim3((unsigned)grid), dim3(512), lds, s, m, n, k, A, lda, B, \
                       ldb, C, ldc, acc, nbm, nbn, deq);                                                              \
  } while (0)
  if constexpr (MFMA_K == 64) {
    if (deq) {
      if (c_fast) MMH_PP_LAUNCH(false, true);
      else MMH_PP_LAUNCH(true, true);
    } else {
      if (c_fast) MMH_PP_LAUNCH(false, false);
      else MMH_PP_LAUNCH(true, false);
    }
  } else {   // the config-named instruction: a forced mode of mmh_igemm_s8 only (no dequantising epilogue)
    if (c_fast) MMH_PP_LAUNCH(false, false);
    else MMH_PP_LAUNCH(true, false);
  }
#undef MMH_PP_LAUNCH
  return hipGetLastError();
}

}  // namespace mmh
