// sgemm_dma.hpp -- K2L: the small-tile SGEMM kernel fed entirely by LDS-DMA.
//
// Why it exists.  Shapes at the small end of the reference sweep (cuda/parameters.h:5-7, N = 1024 ..
// 1792) have fewer 128x128 tiles than the chip has CUs, so they run on 64x64 (or 128x64) tiles -- and
// on those tiles the register-staged packing stage of sgemm_tile.hpp is what bounds the kernel, not
// the matrix pipe.  Timing-only ablations of the 64x64 configuration at 4096^3 (steady state,
// profiles/r02_ablation.md): full kernel 90.8 TFLOP/s, without the global loads 105.8, without the
// registers -> LDS stores as well 143.4, MFMAs alone 149.9.  A 64x64 tile stores 64 KiB into LDS per
// 128-deep K-slice against 4096 matrix-pipe cycles; `ds_write_b128` moves 79 B/clk/CU at best
// (MI355X_MICROARCH.md, LDS) and takes its data through the VGPR ports the MFMAs read their operands
// from -- and the transposing A store needs 16 v_mov per 4x4 block on top.
//
// What it does instead.  BOTH operands go global -> LDS with `buffer_load_dwordx4 ... lds`: no staging
// registers, no ds_write, no v_mov.  The DMA writes lane-linear (lane L -> 16 bytes at base + 16 L), so
// each image is simply the memory layout of its K-slice:
//   As[m][k]  ROW-major (k contiguous, KB floats per row) -- NOT the k-major image of the other
//             kernels: nothing can transpose on the way in.  The MFMA A operand of lane (i = l & 15,
//             kq = l >> 4) for k-step ks is the single float As[m0 + i][4 ks + kq]; the tiles of a
//             wave are 16 rows apart, so one `ds_read2st64_b32` fetches two of them.  The 16-byte chunk
//             index of row m is XORed with (m & 7) -- on the SOURCE side of the DMA (lane L fetches the
//             chunk that belongs at its position) and in the read address -- so that the sixteen rows a
//             read touches do not all sit on one bank (rows are a multiple of 256 B apart); what is left
//             is a 2-way conflict (lanes 0-31 of a ds_read_b32 carry kq = 0, 1 only and so can reach 16
//             of the 32 dword banks), 8 LDS cycles per read instead of 4;
//   Bs[k][n]  as in the other kernels (BN floats per k-row); with 8-byte fragments (wave tiles 32
//             columns wide) odd k-rows have their two halves swapped, again on the source side.
// A ring of NBUF = 3 K-slice buffers: while slice kt is consumed, the DMA pieces of slice kt + NBUF - 1
// are dealt out between its MFMAs (the buffer they land in was last read in slice kt - 1, and every
// wave passed the barrier that ended that slice), and the wait before the barrier that ends slice kt is
// a COUNTED `s_waitcnt vmcnt((NBUF - 2) * pieces)` -- never 0 in the steady state, the newest slices stay
// in flight across it (cdna guide section 5, "what does break it").  Past the last slice the same DMA
// instructions run against zero-length descriptors, so the loop body has no branches.
// All product instantiations use 32-deep slices (KB = 32): rings of 48 KiB (64x64), 72 KiB (128x64) and
// 96 KiB (128x128), so three / two / one workgroup(s) share a CU and one workgroup's pipeline fill,
// hand-over wait or epilogue runs under another's loop (64-deep slices, one workgroup per CU, were 2-7 %
// slower at N <= 1664, profiles/r02_ablation.md section 2).
//
// Arithmetic: the same MFMA (v_mfma_f32_16x16x4_f32) fed the same k's in ascending order as every
// other kernel here -- one fp32 fmaf chain per C element, bit-identical results.
//
// Any shape (EDGE instantiations, round 3).  Nothing can be masked on its way into LDS, so the edges are
// handled where the hardware and the fragment reads allow it:
//   * M / N edges: the descriptors' extents end at the block's last valid A row / B column (the trick of
//     the guarded register-staged kernel, sgemm_mfma.hpp): the per-dword range check turns rows >= m of A
//     and rows >= k of B into zeros IN LDS; columns >= n of B are neighbouring memory and only feed C
//     columns that are never stored;
//   * K tail: A's columns >= k of the last, partial slice are the next row (or the caller's padding) and
//     may be anything, NaN included.  That slice is peeled out of the unrolled ring and run by a copy of
//     the slice body that zeroes the A fragments whose k index is past the end (one v_cndmask per fragment
//     element; B's rows there are zeros already) -- one slice per tile pays the run-time ring index;
//   * 4-byte aligned operands / odd leading dimensions: `buffer_load_dwordx4 ... lds` takes any dword-
//     aligned source (tools/probes/lds_dma_align_probe.hip); C goes out through `aligned(4)` vectors.
// The whole-tile 16-byte-aligned instantiations (EDGE = false) are the round-2 kernels, unchanged.
//
// What was tried and dropped (profiles/r02_ablation.md): four ring buffers (no faster); a run-time ring
// index instead of the unrolled ring (-17 %: an address v_add per fragment read); 8-wave tiles, 256x128
// with three buffers and 256x256 with two -- under the 256-register budget they need the run-time
// ring and then trail the register-staged 256x256 kernel (143-147 vs 149-150 TFLOP/s at N >= 4096);
// the 64x64 tile as EIGHT waves of 16x32 (two waves per SIMD from one workgroup, for tile counts near one
// per CU): 108.2 vs 108.2 TFLOP/s at N=1024 -- a short launch's K loop runs at 73-78 % of the matrix
// pipe's rate either way (tools/dma_timeline.py), the LDS-DMA pieces cost what they cost whoever issues them.
#pragma once
#include <type_traits>

#include "sgemm_tile.hpp"

namespace mmh {

// Timeline stamps for tools/dma_timeline.py (its own build, -DMMH_DMA_TIMELINE, only): workgroup b writes
// the wall clock at kernel entry, after the prologue's barrier, after its K loop and after its C stores
// have completed.
#ifdef MMH_DMA_TIMELINE
__device__ unsigned long long *g_dma_stamps = nullptr;
__device__ int g_dma_stamp_stride = 4;   // 32: the stream-K layout (tools/sk_timeline.py), four slots per part of a range
__device__ __forceinline__ void dma_stamp_value(int i, unsigned long long v) {
  if (g_dma_stamps && threadIdx.x == 0) {
    const int stride = g_dma_stamp_stride;
    g_dma_stamps[(size_t)blockIdx.x * stride + i] = v;
  }
}
__device__ __forceinline__ void dma_stamp(int i) { dma_stamp_value(i, wall_clock64()); }
__device__ __forceinline__ void dma_stamp_after_stores(int i) {
  if (g_dma_stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dma_stamp(i);
  }
}
#else
__device__ __forceinline__ void dma_stamp(int) {}
__device__ __forceinline__ void dma_stamp_after_stores(int) {}
__device__ __forceinline__ void dma_stamp_value(int, unsigned long long) {}
#endif

// One 1 KiB LDS-DMA piece: lane L's 16 bytes land at dst + 16 L.  (A static member of a class, like
// LdsDma in igemm_s8.hpp: buffer descriptors in the signature of a function template trip hipcc's host pass.)
struct DmaPiece {
  static __device__ __forceinline__ void one(__amdgpu_buffer_rsrc_t rsrc, float *dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
  }
};

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
struct DmaTile {
  static_assert((WTM == 1 || WTM == 2 || WTM == 4) && (WTN == 2 || WTN == 4), "wave tile is 16|32|64 x 32|64 (16: measured, not shipped)");
  static_assert(KB == 32 || KB == 64 || KB == 128, "a K-slice row of A is 128, 256 or 512 bytes");
  static_assert(BN == 64 || BN == 128 || BN == 256, "a k-row of B is 256, 512 or 1024 bytes");
  static_assert(NBUF == 3, "a ring of three K-slice buffers");
  static constexpr int WAVES_M = BM / (16 * WTM), WAVES_N = BN / (16 * WTN), WAVES = WAVES_M * WAVES_N;
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int A_FLOATS = BM * KB, B_FLOATS = KB * BN, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int KS = KB / 4;
  static constexpr int CHA = A_FLOATS / 256, CHB = B_FLOATS / 256;           // 1 KiB pieces per image
  static constexpr int CA = CHA / WAVES, CB = CHB / WAVES, ND = CA + CB;     // pieces per wave and slice
  static constexpr int RPC_A = 256 / KB, LPR_A = KB / 4;                     // rows per piece, lanes per row
  static constexpr int RPC_B = 256 / BN, LPR_B = BN / 4;
  static constexpr int LA = NBUF - 1;                                        // slices of look-ahead
  static_assert(CHA % WAVES == 0 && CHB % WAVES == 0 && CA >= 1 && CB >= 1, "pieces divide over the waves");
  static_assert((LA - 1) * ND <= 63, "vmcnt is a 6-bit counter");
  static constexpr size_t LDS_BYTES = (size_t)NBUF * STAGE * sizeof(float);
};

// One C tile (tm, tn), K-slices [kb, ke) of it; same contract as mfma_tile_segment (sgemm_mfma.hpp).
// (A static member of a class template, not a function template: hipcc's host pass mishandles function
// templates whose bodies hold buffer descriptors in generic lambdas -- igemm_s8.hpp has the same note --
// and the 8-wave instantiations then fail to resolve from a second __global__ template.)
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool PART_WT = false, bool EDGE = false>
struct DmaSegment {
static __device__ __forceinline__ void run(float *lds, int m, int n, int k, const float *__restrict__ A, int lda,
                                           const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                           int tm, int tn, int kb, int ke, bool init_from_c,
                                           const float *part_in = nullptr, float *part_out = nullptr,
                                           const SplitFix fix = SplitFix{}) {
  using T = DmaTile<BM, BN, KB, WTM, WTN, NBUF>;
  constexpr int KS = T::KS, STAGE = T::STAGE, A_FLOATS = T::A_FLOATS, CA = T::CA, CB = T::CB, ND = T::ND, LA = T::LA;
  typedef float bfrag_t __attribute__((ext_vector_type(WTN)));
  typedef float afrag_t __attribute__((ext_vector_type(WTM)));
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
  const int li = lane & 15, kq = lane >> 4;
  // C rows / columns of this lane: row(t, r) = crow + 16 t + r, columns ccol .. ccol + WTN - 1
  const int crow = row0 + wm * 16 * WTM + 4 * kq;
  const int ccol = col0 + wn * 16 * WTN + WTN * li;

  // EDGE: the block may hang over the matrix; C may be only 4-byte aligned (the type must say so)
  const int rows_valid = EDGE ? min(BM, m - row0) : BM;
  const int cols_valid = EDGE ? min(BN, n - col0) : BN;
  const bool whole_c = !EDGE || (rows_valid == BM && cols_valid == BN);
  typedef float c_vec_u __attribute__((ext_vector_type(WTN), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, bfrag_t>;

  f32x4 acc[WTM][WTN];
  if (part_in) {
#pragma unroll
    for (int t = 0; t < WTM; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bfrag_t v = *reinterpret_cast<const bfrag_t *>(part_in + (size_t)(crow + 16 * t + r - row0) * BN + (ccol - col0));
#pragma unroll
        for (int u = 0; u < WTN; ++u) acc[t][u][r] = v[u];
      }
  } else if (init_from_c) {
#pragma unroll
    for (int t = 0; t < WTM; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = crow + 16 * t + r;
        bfrag_t v = {};
        if (whole_c) {
          v = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + ccol);
        } else if (row < m) {
#pragma unroll
          for (int u = 0; u < WTN; ++u)
            if (ccol + u < n) v[u] = C[(size_t)row * ldc + ccol + u];
        }
#pragma unroll
        for (int u = 0; u < WTN; ++u) acc[t][u][r] = v[u];
      }
  } else {
#pragma unroll
    for (int t = 0; t < WTM; ++t)
#pragma unroll
      for (int u = 0; u < WTN; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // descriptors: A from (row0, 0), B from (0, col0); a zero-length twin of each for slices past the end
  // (EDGE: extents end at the last valid element of this block's A rows / B columns -- rows >= m of A and
  // rows >= k of B arrive in LDS as zeros)
  const uint32_t ext_a = EDGE ? (uint32_t)(((rows_valid - 1) * lda + k) * 4) : 0x7fffffffu;
  const uint32_t ext_b = EDGE ? (uint32_t)(((k - 1) * ldb + cols_valid) * 4) : 0x7fffffffu;
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A + (size_t)row0 * lda), 0, ext_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B + col0), 0, ext_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, 0, 0x00020000);
  // wave w moves pieces CA w .. CA w + CA - 1 of the A image and likewise of the B image; the 16-byte
  // chunk a lane fetches is the one that belongs at its (swizzled) position
  uint32_t voff_a[CA], voff_b[CB];
#pragma unroll
  for (int j = 0; j < CA; ++j) {
    const int r = T::RPC_A * (CA * wave + j) + lane / T::LPR_A, p = lane % T::LPR_A;
    voff_a[j] = (uint32_t)(r * lda + 4 * (p ^ (r & 7))) * 4u;
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int r = T::RPC_B * (CB * wave + j) + lane / T::LPR_B, p = lane % T::LPR_B;
    voff_b[j] = (uint32_t)(r * ldb + 4 * (WTN == 2 ? (p ^ ((r & 1) << 3)) : p)) * 4u;
  }
  // piece i (0 .. ND-1) of slice kt into ring buffer `buf`
  auto dma_piece = [&](float *buf, int kt, auto i_c) {
    constexpr int I = decltype(i_c)::value;
    const bool live = kt < ke;
    if constexpr (I < CA)
      DmaPiece::one(live ? rsrc_a : null_a, buf + 256 * (CA * wave + I), voff_a[I], (uint32_t)(kt * KB) * 4u);
    else
      DmaPiece::one(live ? rsrc_b : null_b, buf + A_FLOATS + 256 * (CB * wave + (I - CA)), voff_b[I - CA],
                        (uint32_t)(kt * KB) * (uint32_t)ldb * 4u);
  };
  auto dma_slice = [&](float *buf, int kt) { static_for<ND>([&](auto i_c) { dma_piece(buf, kt, i_c); }); };

  // fragment addresses (floats inside a ring buffer): A row-major with the chunk XOR, B as the other kernels
  int a_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a_off[j] = (wm * 16 * WTM + li) * KB + 4 * (j ^ (li & 7)) + kq;
  const int b_off = WTN == 4 ? A_FLOATS + kq * BN + wn * 64 + 4 * li
                             : A_FLOATS + kq * BN + 4 * ((wn * 8 + (li >> 1)) ^ ((kq & 1) << 3)) + 2 * (li & 1);
  auto frag_a = [&](const float *buf, auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    afrag_t a;
#pragma unroll
    for (int t = 0; t < WTM; ++t) a[t] = buf[a_off[ks & 7] + 4 * (ks & ~7) + t * 16 * KB];
    return a;
  };
  auto frag_b = [&](const float *buf, auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    return *reinterpret_cast<const bfrag_t *>(buf + b_off + 4 * ks * BN);
  };

  // ---- prologue: LA slices in flight, the first one landed ----
  static_for<LA>([&](auto s_c) {
    constexpr int S = decltype(s_c)::value;
    dma_slice(lds + S * STAGE, kb + S);
  });
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * ND) : "memory");
  __builtin_amdgcn_s_barrier();
  dma_stamp(1);
  // Fragments are read D = 2 k-steps ahead of the MFMAs that use them, into a ring of four register
  // sets (slot = k-step mod 4; KS is a multiple of 4, so the slot numbering carries over from slice to
  // slice): with four MFMAs (128 matrix-pipe cycles) per k-step one step of distance does not cover an
  // LDS read with a 2-way conflict behind three other waves' reads.
  constexpr int D = 2;
  static_assert(KS % 4 == 0 && KS > 2 * D, "fragment slots are numbered by k-step mod 4");
  afrag_t fa[4];
  bfrag_t fb[4];
  static_for<D>([&](auto d_c) {
    constexpr int d = decltype(d_c)::value;
    fa[d] = frag_a(lds, d_c);
    fb[d] = frag_b(lds, d_c);
  });

  // One K-slice out of ring buffer CUR.  Branch-free and written in issue order (sched_barrier pins it:
  // with so few MFMAs per k-step the order IS the schedule): per k-step the two fragment reads for
  // k-step ks + D, then the MFMAs of k-step ks with at most a few DMA pieces of slice kt + LA behind the
  // first of them (into the buffer slice kt-1 was read from).  The counted wait and the barrier sit
  // before k-step KS - D: from there on the reads go to the NEXT buffer, and every read of this one
  // has been issued by every wave.  The ring index is a compile-time constant (the slice loop below is
  // unrolled over the ring): with a run-time index every fragment read needs a v_add for its address,
  // and beside four MFMAs per k-step that costs 64x64 tiles 17 % (measured: 138 -> 115 TFLOP/s at
  // 4096^3) -- the price is that the compiler renames the accumulators from slice to slice, which is
  // why only the 4-wave tiles (one wave per SIMD, 512 registers) are built this way.
  auto slice = [&](int kt, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % NBUF, DST = (CUR + LA) % NBUF;
    const float *buf = lds + CUR * STAGE;
    const float *nxt = lds + NXT * STAGE;
    float *dst = lds + DST * STAGE;
    static_for<KS>([&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      // which DMA pieces ride on this k-step: ND pieces over k-steps 0 .. KS-D-1
      constexpr int SPAN = KS - D;
      constexpr int P0 = ks < SPAN ? ks * ND / SPAN : ND, P1 = ks < SPAN ? (ks + 1) * ND / SPAN : ND;
      if constexpr (ks == KS - D) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((LA - 1) * ND) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (ks + D < KS) {
        fa[(ks + D) & 3] = frag_a(buf, std::integral_constant<int, ks + D>{});
        fb[(ks + D) & 3] = frag_b(buf, std::integral_constant<int, ks + D>{});
      } else {
        fa[(ks + D) & 3] = frag_a(nxt, std::integral_constant<int, ks + D - KS>{});
        fb[(ks + D) & 3] = frag_b(nxt, std::integral_constant<int, ks + D - KS>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      const afrag_t a = fa[ks & 3];
      const bfrag_t b = fb[ks & 3];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      static_for<P1 - P0>([&](auto i_c) { dma_piece(dst, kt + LA, std::integral_constant<int, P0 + decltype(i_c)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < WTM; ++t)
#pragma unroll
        for (int u = 0; u < WTN; ++u)
          if (t + u > 0) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // EDGE: the problem's last slice when k is not a multiple of KB -- A's fragments past column k are zeroed
  // as they are used (they hold the next row's floats or the caller's padding, NaN included; B's rows there
  // are zeros by descriptor).  Peeled: its ring position is a run-time value, which the steady-state slices
  // must not pay for (see `slice`).  Its fragments for k-steps 0 .. D-1 were read by whatever ran before it.
  const bool ragged_k = EDGE && ke * KB > k;
  const int ke_main = ragged_k ? ke - 1 : ke;
  int kt = kb;
  if (kt < ke_main)
    for (;;) {
      static_assert(NBUF == 3, "the slice loop is unrolled over a ring of three (four was measured: no faster)");
      slice(kt, std::integral_constant<int, 0>{});
      if (++kt >= ke_main) break;
      slice(kt, std::integral_constant<int, 1>{});
      if (++kt >= ke_main) break;
      slice(kt, std::integral_constant<int, 2>{});
      if (++kt >= ke_main) break;
    }
  if constexpr (EDGE) {
    if (ragged_k) {
      const int krem = k - kt * KB;                       // valid k's of this slice, 1 .. KB-1
      const float *buf = lds + ((kt - kb) % NBUF) * STAGE;
      static_for<KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks + D < KS) {
          fa[(ks + D) & 3] = frag_a(buf, std::integral_constant<int, ks + D>{});
          fb[(ks + D) & 3] = frag_b(buf, std::integral_constant<int, ks + D>{});
        }
        afrag_t a = fa[ks & 3];
        bfrag_t b = fb[ks & 3];
        const bool live = 4 * ks + kq < krem;   // this lane's k of the k-step (the same k for its A and its B operand)
#pragma unroll
        for (int t = 0; t < WTM; ++t) a[t] = live ? a[t] : 0.0f;
#pragma unroll
        for (int u = 0; u < WTN; ++u) b[u] = live ? b[u] : 0.0f;   // (zeros by descriptor already; belt and braces)
#pragma unroll
        for (int t = 0; t < WTM; ++t)
#pragma unroll
          for (int u = 0; u < WTN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
      });
    }
  }
  // keep the fragments prefetched past the last slice formally alive: otherwise the compiler sinks the
  // reads that follow each slice's barrier into the NEXT slice's block (they are dead on the exit path)
  // and the first k-step of every slice waits for them
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm volatile("" ::"v"(fb[i]));
#pragma unroll
    for (int t = 0; t < WTM; ++t) asm volatile("" ::"v"(fa[i][t]));   // (element-wise: a 1-vector has no register class)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-length tail DMAs: nothing may still be landing in LDS
  dma_stamp(2);

  // split-K finisher (see mfma_tile_segment): add the other parts' partial tiles in part order
  if (fix.count > 0) {
    int bad = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
      long long spins = 0;
      while (__hip_atomic_load(fix.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fix.count) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > fix.spin_limit) { bad = 1; break; }
      }
      if (bad) __hip_atomic_fetch_add(fix.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (!bad) __hip_atomic_store(fix.flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // last reader: leave it zero
      reinterpret_cast<volatile int *>(lds)[0] = bad;
    }
    __syncthreads();
    if (reinterpret_cast<volatile int *>(lds)[0]) return;
    for (int p = 0; p < fix.count; ++p) {
      const float *src = fix.parts + (size_t)p * fix.stride;
#pragma unroll
      for (int t = 0; t < WTM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bfrag_t v = *reinterpret_cast<const bfrag_t *>(src + (size_t)(crow + 16 * t + r - row0) * BN + (ccol - col0));
#pragma unroll
          for (int u = 0; u < WTN; ++u) acc[t][u][r] += v[u];
        }
    }
  }

  __amdgpu_buffer_rsrc_t rsrc_p;
  if (PART_WT && part_out) rsrc_p = __builtin_amdgcn_make_buffer_rsrc(part_out, 0, BM * BN * 4, 0x00020000);
#pragma unroll
  for (int t = 0; t < WTM; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow + 16 * t + r;
      bfrag_t v;
#pragma unroll
      for (int u = 0; u < WTN; ++u) v[u] = acc[t][u][r];
      if (part_out) {
        if constexpr (PART_WT) {
          const uint32_t off = (uint32_t)(((row - row0) * BN + (ccol - col0)) * 4);
          if constexpr (WTN == 4) {
            typedef int i32x4_t __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsrc_p, off, 0, 16);
          } else {
            typedef int i32x2_t __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, v), rsrc_p, off, 0, 16);
          }
        } else {
          *reinterpret_cast<bfrag_t *>(part_out + (size_t)(row - row0) * BN + (ccol - col0)) = v;
        }
      } else if (whole_c) {
        *reinterpret_cast<c_vec *>(C + (size_t)row * ldc + ccol) = v;
      } else if (row < m) {
#pragma unroll
        for (int u = 0; u < WTN; ++u)
          if (ccol + u < n) C[(size_t)row * ldc + ccol + u] = v[u];
      }
    }
}
};

// One workgroup per C tile (XCD-aware block -> tile map), whole K range.
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE = false>
__global__ void __launch_bounds__((BM / (16 * WTM)) * (BN / (16 * WTN)) * 64)
sgemm_mfma_dma_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                      float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int tm, tn;
  dma_stamp(0);
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  DmaSegment<BM, BN, KB, WTM, WTN, NBUF, false, EDGE>::run(lds, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, 0,
                                                           (k + KB - 1) / KB, accumulate != 0);
  dma_stamp_after_stores(3);
}

// (round 3's rim -- extra workgroups of the same launch computing the strips beyond the last tile boundary on the vector
// ALU; measured, it does not pay: profiles/r03_notes.md section 6 -- lives in tools/ab/sgemm_dma_rim.hpp: tools build only)

// Segment policy of this tile for the chained stream-K control flow (streamk_body in sgemm_mfma.hpp,
// K2p) -- tile counts that do not divide the chip run as one persistent workgroup per CU over ranges
// of (tile, K-slice) units, partial tiles handed over through the workspace: sgemm_dma_streamk_kernel.
template <int BM_, int BN_, int KB_, int WTM, int WTN, int NBUF, bool EDGE = false>
struct DmaSeg {
  static constexpr int BM = BM_, BN = BN_, KB = KB_;
  static constexpr int THREADS = DmaTile<BM_, BN_, KB_, WTM, WTN, NBUF>::THREADS;
  static __device__ __forceinline__ void run(float *lds, int m, int n, int k, const float *__restrict__ A, int lda,
                                             const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                             int tm, int tn, int kb, int ke, bool init_from_c, const float *part_in,
                                             float *part_out) {
    DmaSegment<BM, BN, KB, WTM, WTN, NBUF, true, EDGE>::run(lds, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, kb, ke,
                                                            init_from_c, part_in, part_out);   // partial tiles write-through
  }
};

}  // namespace mmh
