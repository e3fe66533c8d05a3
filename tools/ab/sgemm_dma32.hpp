// sgemm_dma32.hpp -- K2M: the LDS-DMA tile on v_mfma_f32_32x32x2_f32 (round 4).
//
// Why it exists.  K2L (sgemm_dma.hpp) feeds v_mfma_f32_16x16x4_f32: 32 matrix-pipe cycles per instruction, so
// everything else a wave has to issue between two of them -- an LDS-DMA piece (~60 cycles of issue, MI355X_MICROARCH.md,
// "LDS-DMA piece issue cost"), two fragment reads -- has 32 cycles of cover, and with ONE workgroup per CU (every size
// of the reference sweep below N = 1408, cuda/parameters.h:5-7, and every persistent stream-K launch of the big tiles)
// nobody else fills the gap: the K loop of a lone 64x64 workgroup runs at 87 % of the pipe's rate.  The 32x32x2 form
// does the same arithmetic -- D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), bit for bit the chain (cdna guide section 3) --
// in 64-cycle instructions that depend on nothing but their own accumulator (dependent latency = issue interval), so
// a wave needs HALF as many matrix instructions per flop and each one covers twice as much issue time.
//
// Operand shapes.  A operand of lane l: A[i = l & 31][k = l >> 5]; B operand: B[k = l >> 5][j = l & 31]; one float each.
//   As[m][k]  the row-major image of K2L (nothing can transpose on the way in), 32-deep slices, 16-byte chunks XORed
//             with (m >> 1) & 7 on the source side of the DMA.  A lane reads ONE ds_read_b128 per eight k's: lanes
//             0-31 the chunk k = 8g .. 8g+3 of their row, lanes 32-63 the chunk 8g+4 .. 8g+7 of the same rows, and two
//             v_permlane32_swap_b32 turn the four registers {k0|k4} {k1|k5} {k2|k6} {k3|k7} (lower | upper half-wave)
//             into the four operands {k0|k1} {k2|k3} {k4|k5} {k6|k7}: no wasted LDS bytes, no select, and the read is
//             bank-conflict free (the sixteen lanes of a ds_read_b128 service group touch sixteen different 16-byte
//             slots of the 256-byte bank row: odd / even rows take its two halves, the XOR spreads the rest) where
//             K2L's single-float reads pay a 2-way conflict;
//   Bs[k][n]  as it lies in memory, no swizzle: the 32 lanes of a half-wave read WN consecutive floats each of ONE
//             k-row (128 or 256 contiguous bytes), the other half-wave the next k-row.
// Column interleave as in the other kernels: block u of a wave covers columns {n0 + WN j + u}, so a lane holds WN
// consecutive columns of a C row and stores them as one vector.
//
// Schedule.  A K-slice is KG = 4 k-groups of eight k's; per group a wave issues 4 WM WN MFMAs.  Fragments are read one
// group ahead into the other of two register sets; the LDS-DMA pieces of slice kt + 2 are dealt out one per MFMA gap
// over the four groups; the counted wait and the slice's one barrier sit in front of the last group's reads (which
// go to the next ring buffer).  Ring of three 32-deep slices as K2L: 48 / 72 / 96 KiB = 3 / 2 / 1 workgroups per CU.
//
// Chained segments (CHAIN).  A persistent stream-K workgroup runs several (tile, K-range) segments back to back.  K2L
// starts each with an empty pipeline: two slices of DMA latency with nothing to overlap (one workgroup per CU), and the
// previous segment's C / partial-tile stores drain first.  Here the slices of consecutive segments form ONE stream
// through the ring: while a segment's last two slices are consumed, the DMA instructions that K2L issues against
// zero-length descriptors fetch the NEXT segment's first two slices, the fragment reads at the end of the last slice
// are the next segment's first, and the epilogue's stores run under loads already in flight.  The ring position a
// segment starts at is then a run-time value: up to two slices run from copies of the slice body that sit in front of
// the unrolled ring loop (the steady state keeps its compile-time LDS offsets).
#pragma once
#include <type_traits>

#include "sgemm_dma.hpp"
#include "sgemm_mfma.hpp"   // the stream-K hand-over protocol (SK_* words), streamk_body

namespace mmh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

// MB = row blocks per matrix instruction: 1 = v_mfma_f32_32x32x2_f32 (one 32x32 block, two k's), 2 =
// v_mfma_f32_32x32x1_2b_f32 (two 32x32 blocks -- 64 rows -- one k: lanes 0-31 carry block 0's A rows, lanes 32-63
// block 1's, BOTH at the same k, so the four registers of a lane's ds_read_b128 are the A operands of four consecutive
// instructions as they stand: no v_permlane32_swap, which measured 9 % of the MB = 1 loop, profiles/r04_notes.md).
template <int BM, int BN, int KB, int WM, int WN, int NBUF, int MB = 1>
struct Dma32Tile {
  static_assert(KB == 32, "a K-slice row of A is 128 bytes: eight 16-byte chunks");
  static_assert(MB == 1 || MB == 2, "one or two 32-row blocks per matrix instruction");
  static_assert(WM >= 1 && WM <= 4 && (WN == 1 || WN == 2 || WN == 4), "wave tile is 32 MB WM x 32 WN");
  static_assert(NBUF == 3, "a ring of three K-slice buffers");
  static constexpr int RB = 32 * MB;                                          // rows per matrix instruction
  static constexpr int WAVES_M = BM / (RB * WM), WAVES_N = BN / (32 * WN), WAVES = WAVES_M * WAVES_N;
  static_assert(WAVES_M * RB * WM == BM && WAVES_N * 32 * WN == BN && WAVES == 4, "four waves, one per SIMD");
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int A_FLOATS = BM * KB, B_FLOATS = KB * BN, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int KSTEP = 8 / MB;                                        // k's per k-group: four matrix instructions deep
  static constexpr int KG = KB / KSTEP;                                       // k-groups per slice
  static constexpr int CHA = A_FLOATS / 256, CHB = B_FLOATS / 256;            // 1 KiB pieces per image
  static constexpr int CA = CHA / WAVES, CB = CHB / WAVES, ND = CA + CB;      // pieces per wave and slice
  static constexpr int RPC_A = 256 / KB, LPR_A = KB / 4;                      // rows per piece, lanes per row
  static constexpr int RPC_B = 256 / BN > 0 ? 256 / BN : 1, LPR_B = BN / 4;
  static constexpr int LA = NBUF - 1;                                         // slices of look-ahead
  static_assert(CHA % WAVES == 0 && CHB % WAVES == 0 && CA >= 1 && CB >= 1, "pieces divide over the waves");
  static_assert(BN <= 256, "a piece holds whole k-rows of B");
  static_assert(LA * ND <= 63, "vmcnt is a 6-bit counter");
  // the ring, plus one 64-byte line the chained stream-K body passes a word between its waves through
  static constexpr size_t RING_BYTES = (size_t)NBUF * STAGE * sizeof(float);
  static constexpr size_t LDS_BYTES = RING_BYTES + 64;
};

// What a segment needs to know about the one that follows it in the workgroup's stream (CHAIN): whose first LA slices
// the tail's look-ahead fetches.  `len` = its number of slices; len < LA (or no successor: len = 0) breaks the chain --
// the tail runs against empty descriptors as in K2L and the next segment starts with a prologue of its own.
struct Dma32Next {
  int tm = 0, tn = 0, kb = 0, len = 0;
};
// The state a chained workgroup carries from segment to segment: the ring position the next slice of the stream lands
// in, and whether that segment's first slices (and first fragments) are already on their way.
struct Dma32Link {
  int pos = 0;
  bool primed = false;
};

// ABL (tools build only, timing-only: WRONG results): 1 no swaps, 2 no LDS-DMA inside the loop, 4 no A fragment
// reads, 8 no B fragment reads -- what each part of the loop costs (profiles/r04_notes.md).
template <int BM, int BN, int KB, int WM, int WN, int NBUF, bool PART_WT = false, bool EDGE = false, bool CHAIN = false, int ABL = 0,
          int MB = 1>
struct Dma32Segment {
  using T = Dma32Tile<BM, BN, KB, WM, WN, NBUF, MB>;
  static constexpr int RB = T::RB, KSTEP = T::KSTEP;
  using acc_t = std::conditional_t<MB == 2, f32x32, f32x16>;
  typedef float bvec_t __attribute__((ext_vector_type(WN)));
  typedef float c_vec_u __attribute__((ext_vector_type(WN), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, bvec_t>;

  // fragment registers of one k-group: the A operands of its four k-pairs per row block (after the swaps), and the
  // four k-pairs' B vectors
  struct Frag {
    float a[WM][4];
    bvec_t b[4];
  };

  // The per-lane constants of a workgroup (independent of the tile): computed once per kernel.
  struct Lane {
    int wave, wm, wn, i, h;
    int a_off[T::KG];   // floats inside a ring buffer: this lane's A chunk of k-group g (row block 0)
    int b_off;          // ... and its B vector of k-pair 0 of group 0
    uint32_t voff_a[T::CA], voff_b[T::CB];
    __device__ __forceinline__ void init(int lda, int ldb) {
      const int tid = threadIdx.x, lane = tid & 63;
      wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      wm = wave / T::WAVES_N;
      wn = wave % T::WAVES_N;
      i = lane & 31;
      h = lane >> 5;
      const int sw = (i >> 1) & 7;
#pragma unroll
      for (int g = 0; g < T::KG; ++g)
        a_off[g] = MB == 1 ? (wm * RB * WM + i) * KB + 4 * ((2 * g + h) ^ sw)       // lower / upper half-wave: chunks 2g / 2g+1
                           : (wm * RB * WM + 32 * h + i) * KB + 4 * (g ^ sw);       // ... rows i / 32 + i of the same chunk
      b_off = T::A_FLOATS + (MB == 1 ? h * BN : 0) + wn * 32 * WN + WN * i;
      // wave w moves pieces CA w .. CA w + CA - 1 of the A image and likewise of the B image; the 16-byte chunk a
      // lane fetches is the one that belongs at its (swizzled) position
#pragma unroll
      for (int j = 0; j < T::CA; ++j) {
        const int r = T::RPC_A * (T::CA * wave + j) + lane / T::LPR_A, p = lane % T::LPR_A;
        voff_a[j] = (uint32_t)(r * lda + 4 * (p ^ ((r >> 1) & 7))) * 4u;
      }
#pragma unroll
      for (int j = 0; j < T::CB; ++j) {
        const int q = 64 * (T::CB * wave + j) + lane;   // 16-byte chunk index inside the B image
        voff_b[j] = (uint32_t)((q / T::LPR_B) * ldb + 4 * (q % T::LPR_B)) * 4u;
      }
    }
  };

  template <int g>
  static __device__ __forceinline__ void read_frag(Frag &f, const float *buf, const Lane &L) {
#pragma unroll
    for (int bm = 0; bm < ((ABL & 4) ? 0 : WM); ++bm) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + L.a_off[g] + bm * RB * KB);
#pragma unroll
      for (int q = 0; q < 4; ++q) f.a[bm][q] = v[q];
    }
#pragma unroll
    for (int p = 0; p < ((ABL & 8) ? 0 : 4); ++p) f.b[p] = *reinterpret_cast<const bvec_t *>(buf + L.b_off + (KSTEP * g + (2 / MB) * p) * BN);
  }
  // {k0|k4} {k1|k5} {k2|k6} {k3|k7}  ->  a[0] = {k0|k1}, a[1] = {k2|k3}, a[2] = {k4|k5}, a[3] = {k6|k7}
  static __device__ __forceinline__ void swap_frag(Frag &f) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int bm = 0; bm < ((ABL & 1) || MB == 2 ? 0 : WM); ++bm) {
      const u32x2 s01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, f.a[bm][0]),
                                                         __builtin_bit_cast(unsigned, f.a[bm][1]), false, false);
      const u32x2 s23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, f.a[bm][2]),
                                                         __builtin_bit_cast(unsigned, f.a[bm][3]), false, false);
      // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index -- clang 19)
      const unsigned k01 = s01[0], k45 = s01[1], k23 = s23[0], k67 = s23[1];
      f.a[bm][0] = __builtin_bit_cast(float, k01);
      f.a[bm][1] = __builtin_bit_cast(float, k23);
      f.a[bm][2] = __builtin_bit_cast(float, k45);
      f.a[bm][3] = __builtin_bit_cast(float, k67);
    }
  }

  // One C tile (tm, tn), K-slices [kb, ke) of it; the contract of DmaSegment::run (sgemm_dma.hpp).  CHAIN adds `nx`
  // (the segment that follows in this workgroup's stream) and `link` (in: where this segment's slices start in the
  // ring and whether they are already in flight; out: the same for the next one).
  static __device__ __forceinline__ void run(float *lds, const Lane &L, int m, int n, int k, const float *__restrict__ A,
                                             int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                             int tm, int tn, int kb, int ke, bool init_from_c,
                                             const float *part_in, float *part_out, Frag (&fr)[2], Dma32Link &link,
                                             const Dma32Next nx = Dma32Next{}) {
    constexpr int KG = T::KG, STAGE = T::STAGE, A_FLOATS = T::A_FLOATS, CA = T::CA, CB = T::CB, ND = T::ND, LA = T::LA;
    const int row0 = tm * BM, col0 = tn * BN;
    // C rows / columns of this lane: row(bm, r) = crow + 32 bm + (r & 3) + 8 (r >> 2), columns ccol .. ccol + WN - 1
    const int crow = row0 + L.wm * RB * WM + 4 * L.h;
    const int ccol = col0 + L.wn * 32 * WN + WN * L.i;
    const int rows_valid = EDGE ? min(BM, m - row0) : BM;
    const int cols_valid = EDGE ? min(BN, n - col0) : BN;
    const bool whole_c = !EDGE || (rows_valid == BM && cols_valid == BN);

    // descriptors: A from (row0, 0), B from (0, col0) (EDGE: extents end at the last valid element of this block's A
    // rows / B columns -- rows >= m of A and rows >= k of B arrive in LDS as zeros); the successor's likewise, or empty
    // ones (as in K2L: past the last slice the same DMA instructions run against zero-length descriptors).  Kept as
    // base + extent SCALARS and packed where they are used: a select between two 128-bit descriptors sends hipcc
    // through scratch memory and a waterfall loop.
    auto ext_a = [&](int valid) { return EDGE ? (uint32_t)(((valid - 1) * lda + k) * 4) : 0x7fffffffu; };
    auto ext_b = [&](int valid) { return EDGE ? (uint32_t)(((k - 1) * ldb + valid) * 4) : 0x7fffffffu; };
    const float *own_pa = A + (size_t)row0 * lda, *own_pb = B + col0;
    const uint32_t own_ea = ext_a(rows_valid), own_eb = ext_b(cols_valid);
    const bool chain = CHAIN && nx.len >= LA;
    const float *next_pa = A, *next_pb = B;
    uint32_t next_ea = 0, next_eb = 0;
    if constexpr (CHAIN) {
      if (chain) {
        next_pa = A + (size_t)(nx.tm * BM) * lda;
        next_pb = B + nx.tn * BN;
        next_ea = ext_a(EDGE ? min(BM, m - nx.tm * BM) : BM);
        next_eb = ext_b(EDGE ? min(BN, n - nx.tn * BN) : BN);
      }
    }
    const int kdelta = nx.kb - ke;   // stream slice s >= ke of this segment is slice s + kdelta of the next
    // the descriptors and slice offset of stream slice kt
    struct Src {
      __amdgpu_buffer_rsrc_t a, b;
      uint32_t off_a, off_b;
    };
    auto source = [&](int kt) {
      const bool own = kt < ke;
      const int ks = own ? kt : kt + kdelta;
      Src s;
      s.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(own ? own_pa : next_pa), 0, own ? own_ea : next_ea, 0x00020000);
      s.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(own ? own_pb : next_pb), 0, own ? own_eb : next_eb, 0x00020000);
      s.off_a = (uint32_t)(ks * KB) * 4u;
      s.off_b = (uint32_t)(ks * KB) * (uint32_t)ldb * 4u;
      return s;
    };
    // piece I (0 .. ND-1) of a stream slice into ring buffer `buf`
    auto dma_piece = [&](float *buf, const Src &s, auto i_c) {
      constexpr int I = decltype(i_c)::value;
      if constexpr (I < CA)
        DmaPiece::one(s.a, buf + 256 * (CA * L.wave + I), L.voff_a[I], s.off_a);
      else
        DmaPiece::one(s.b, buf + A_FLOATS + 256 * (CB * L.wave + (I - CA)), L.voff_b[I - CA], s.off_b);
    };

    acc_t acc[WM][WN];
    constexpr int NR = 16 * MB;   // accumulator registers per instruction: register r = row 32 (r >> 4) + (r & 3) + 8 ((r & 15) >> 2) + 4 h
    auto c_row = [&](int bm, int r) { return crow + RB * bm + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2); };
    // ---- prologue: LA slices in flight, the first one landed, its first fragments read (unless the previous
    // segment of the chain has done all that) ----
    int pos = CHAIN ? __builtin_amdgcn_readfirstlane(link.pos) : 0;
    const bool primed = CHAIN && __builtin_amdgcn_readfirstlane((int)link.primed) != 0;
    if (!primed) {
      if constexpr (CHAIN) __syncthreads();   // every wave is past its last fragment read of whatever ran before
      pos = 0;
      static_for<LA>([&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        const Src src = source(kb + S);
        static_for<ND>([&](auto i_c) { dma_piece(lds + S * STAGE, src, i_c); });
      });
    }
    // the accumulators (their loads, if any, run under the DMA latency)
    if (part_in) {
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int rr = c_row(bm, r) - row0;
          const bvec_t v = *reinterpret_cast<const bvec_t *>(part_in + (size_t)rr * BN + (ccol - col0));
#pragma unroll
          for (int u = 0; u < WN; ++u) acc[bm][u][r] = v[u];
        }
    } else if (init_from_c) {
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = c_row(bm, r);
          bvec_t v = {};
          if (whole_c) {
            v = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + ccol);
          } else if (row < m) {
#pragma unroll
            for (int u = 0; u < WN; ++u)
              if (ccol + u < n) v[u] = C[(size_t)row * ldc + ccol + u];
          }
#pragma unroll
          for (int u = 0; u < WN; ++u) acc[bm][u][r] = v[u];
        }
    } else {
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int u = 0; u < WN; ++u)
#pragma unroll
          for (int r = 0; r < NR; ++r) acc[bm][u][r] = 0.0f;
    }
    if (!primed) {
      if (part_in || init_from_c) {
        // the accumulators' loads were issued AFTER the DMAs: waiting for the first slice means waiting for them too
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * ND) : "memory");
      }
      __builtin_amdgcn_s_barrier();
      read_frag<0>(fr[0], lds, L);
      swap_frag(fr[0]);
    } else if (part_in || init_from_c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (chained: the stream's slices in flight land as well -- a tail's price)
    }
    dma_stamp(1);

    // One K-slice out of ring buffer CUR.  Branch-free and written in issue order.  Per k-group g: the fragment reads
    // of group g + 1 (the last group's go to the NEXT ring buffer, behind the counted wait and the slice's barrier),
    // then the group's 4 WM WN MFMAs with the group's share of the DMA pieces of stream slice kt + LA behind the first
    // ones (into the buffer slice kt - 1 was read from), the swaps of the fragments just read before the last k-pair.
    auto slice_at = [&](int kt, const float *buf, const float *nxt, float *dst, auto tail_c) {
      constexpr bool TAIL = decltype(tail_c)::value;   // EDGE: the problem's last, partial slice (see below)
      const int krem = TAIL ? k - kt * KB : KB;
      const Src src = source(kt + LA);
      static_for<KG>([&](auto g_c) {
        constexpr int g = decltype(g_c)::value;
        constexpr int P0 = g * ND / KG, P1 = (g + 1) * ND / KG;   // this group's DMA pieces
        Frag &cur = fr[g & 1];
        Frag &nxf = fr[(g + 1) & 1];
        if constexpr (g == KG - 1) {
          // everything but the pieces of slice kt + LA issued so far in this slice has landed: slice kt + 1 is whole
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((ABL & 2) ? 0 : (KG - 1) * ND / KG) : "memory");
          __builtin_amdgcn_s_barrier();
          read_frag<0>(nxf, nxt, L);
        } else {
          read_frag<(g + 1) % KG>(nxf, buf, L);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAIL) {
          // A's columns past k hold the next row's floats or the caller's padding (NaN included): zero this lane's
          // operands of the k's that do not exist (B's rows there are zeros by descriptor; belt and braces)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const bool live = KSTEP * g + (MB == 1 ? 2 * p + L.h : p) < krem;
#pragma unroll
            for (int bm = 0; bm < WM; ++bm) cur.a[bm][p] = live ? cur.a[bm][p] : 0.0f;
#pragma unroll
            for (int u = 0; u < WN; ++u) cur.b[p][u] = live ? cur.b[p][u] : 0.0f;
          }
        }
        static_for<4>([&](auto p_c) {
          constexpr int p = decltype(p_c)::value;
          if constexpr (p == 3) {
            swap_frag(nxf);
            __builtin_amdgcn_sched_barrier(0);
          }
          static_for<WM * WN>([&](auto q_c) {
            constexpr int q = decltype(q_c)::value, bm = q / WN, u = q % WN;
            if constexpr (MB == 1) acc[bm][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[bm][p], cur.b[p][u], acc[bm][u], 0, 0, 0);
            else acc[bm][u] = __builtin_amdgcn_mfma_f32_32x32x1f32(cur.a[bm][p], cur.b[p][u], acc[bm][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // one DMA piece behind each of the group's first MFMAs
            constexpr int idx = p * WM * WN + q;
            if constexpr (idx < P1 - P0 && !(ABL & 2)) {
              dma_piece(dst, src, std::integral_constant<int, P0 + idx>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          });
        });
      });
    };
    // the steady state: ring position CUR at compile time (LDS offsets become immediates)
    auto slice = [&](int kt, auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % NBUF, DST = (CUR + LA) % NBUF;
      slice_at(kt, lds + CUR * STAGE, lds + NXT * STAGE, lds + DST * STAGE, std::false_type{});
    };

    // EDGE: the problem's last slice when k is not a multiple of KB runs the TAIL copy of the slice body.
    const bool ragged_k = EDGE && ke * KB > k;
    const int ke_main = ragged_k ? ke - 1 : ke;
    int kt = kb;
    // The main part.  ONE exit per loop: with `break`s between the unrolled slices the accumulators of the exits meet
    // in different registers and hipcc copies all of them (v_accvgpr_mov) on the hot path of every slice.
    if constexpr (CHAIN) {   // up to two slices to reach ring position 0
      if (pos == 1 && kt < ke_main) { slice(kt, std::integral_constant<int, 1>{}); ++kt; pos = 2; }
      if (pos == 2 && kt < ke_main) { slice(kt, std::integral_constant<int, 2>{}); ++kt; pos = 0; }
    }
    while (kt + NBUF <= ke_main) {
      slice(kt, std::integral_constant<int, 0>{});
      slice(kt + 1, std::integral_constant<int, 1>{});
      slice(kt + 2, std::integral_constant<int, 2>{});
      kt += NBUF;
    }
    if (kt < ke_main) {   // (only reached at ring position 0)
      slice(kt, std::integral_constant<int, 0>{});
      ++kt;
      pos = 1;
      if (kt < ke_main) {
        slice(kt, std::integral_constant<int, 1>{});
        ++kt;
        pos = 2;
      }
    }
    if constexpr (EDGE) {
      if (ragged_k) {
        // (one slice per tile pays for a run-time ring position: an address add per fragment read)
        const int nx1 = pos == 2 ? 0 : pos + 1, nx2 = nx1 == 2 ? 0 : nx1 + 1;
        slice_at(kt, lds + pos * STAGE, lds + nx1 * STAGE, lds + nx2 * STAGE, std::true_type{});
        pos = nx1;
      }
    }
    if constexpr (CHAIN) {
      link.pos = pos;
      link.primed = chain;
    }
    if (!CHAIN || !chain) {
      // keep the fragments prefetched past the last slice formally alive (see sgemm_dma.hpp), and let the tail's
      // DMAs against the empty descriptors finish: nothing may still be landing in LDS
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int u = 0; u < WN; ++u) asm volatile("" ::"v"(fr[s].b[p][u]));   // (element-wise: a 1-vector has no register class)
#pragma unroll
          for (int bm = 0; bm < WM; ++bm) asm volatile("" ::"v"(fr[s].a[bm][p]));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    dma_stamp(2);

    // the tile goes out: a partial tile into the workspace (dense BM x BN, write-through), or C (whole tiles as
    // vectors, edge tiles element by element).  (One branch around each store loop, not one inside per store.)
    auto out_vec = [&](int bm, int r) {
      bvec_t v;
#pragma unroll
      for (int u = 0; u < WN; ++u) v[u] = acc[bm][u][r];
      return v;
    };
    if (part_out) {
      __amdgpu_buffer_rsrc_t rsrc_p;
      if constexpr (PART_WT) rsrc_p = __builtin_amdgcn_make_buffer_rsrc(part_out, 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = c_row(bm, r);
          const bvec_t v = out_vec(bm, r);
          if constexpr (PART_WT) {
            const uint32_t off = (uint32_t)(((row - row0) * BN + (ccol - col0)) * 4);
            if constexpr (WN == 4) {
              typedef int i32x4_t __attribute__((ext_vector_type(4)));
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsrc_p, off, 0, 16);
            } else if constexpr (WN == 2) {
              typedef int i32x2_t __attribute__((ext_vector_type(2)));
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, v), rsrc_p, off, 0, 16);
            } else {
              const float v0 = v[0];
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v0), rsrc_p, off, 0, 16);
            }
          } else {
            *reinterpret_cast<bvec_t *>(part_out + (size_t)(row - row0) * BN + (ccol - col0)) = v;
          }
        }
    } else if (whole_c) {
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int r = 0; r < NR; ++r) *reinterpret_cast<c_vec *>(C + (size_t)c_row(bm, r) * ldc + ccol) = out_vec(bm, r);
    } else {
#pragma unroll
      for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = c_row(bm, r);
          const bvec_t v = out_vec(bm, r);
          if (row < m) {
#pragma unroll
            for (int u = 0; u < WN; ++u)
              if (ccol + u < n) C[(size_t)row * ldc + ccol + u] = v[u];
          }
        }
    }
  }
};

// One workgroup per C tile (XCD-aware block -> tile map), whole K range.
template <int BM, int BN, int KB, int WM, int WN, int NBUF, bool EDGE = false, int ABL = 0, int MB = 1>
__global__ void __launch_bounds__(256)
sgemm_mfma32_dma_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                        float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = Dma32Segment<BM, BN, KB, WM, WN, NBUF, false, EDGE, false, ABL, MB>;
  int tm, tn;
  dma_stamp(0);
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  typename S::Lane L;
  L.init(lda, ldb);
  typename S::Frag fr[2];
  Dma32Link link;
  S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, 0, (k + KB - 1) / KB, accumulate != 0, nullptr, nullptr, fr, link);
  dma_stamp_after_stores(3);
}

// Segment policy for streamk_body (sgemm_mfma.hpp): the UNCHAINED form, every segment with its own prologue.
template <int BM_, int BN_, int KB_, int WM, int WN, int NBUF, bool EDGE = false, int MB = 1>
struct Dma32Seg {
  static constexpr int BM = BM_, BN = BN_, KB = KB_;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(float *lds, int m, int n, int k, const float *__restrict__ A, int lda,
                                             const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                             int tm, int tn, int kb, int ke, bool init_from_c, const float *part_in,
                                             float *part_out) {
    using S = Dma32Segment<BM, BN, KB, WM, WN, NBUF, true, EDGE, false, 0, MB>;
    typename S::Lane L;
    L.init(lda, ldb);
    typename S::Frag fr[2];
    Dma32Link link;
    S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, kb, ke, init_from_c, part_in, part_out, fr, link);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// K2Mp: the chained stream-K body.  The ranges, the order of a range's parts (head of the last tile FIRST, whole tiles,
// tail of the first tile LAST), the hand-over protocol and its words are streamk_body's (sgemm_mfma.hpp, K2p) -- a
// launch of either body interoperates with the same workspace and tables.  What differs is how the parts run: as ONE
// stream of K-slices through the ring (Dma32Segment, CHAIN): each part's tail fetches the next part's first two
// slices, so that only the workgroup's first part pays a pipeline fill, and the stores of a part's C tile (or of the
// partial tile, and the wait for them that precedes the publish) run under loads already in flight.  A tail's first
// slices are fetched BEFORE its hand-over word is looked at: they depend on nobody; should the word say the head's
// owner is not running (the wait-free path: leave), they are simply dropped.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KB, int WM, int WN, int NBUF, bool EDGE, int MB>
__device__ __forceinline__ void streamk32_body(float *lds, int m, int n, int k, const float *__restrict__ A, int lda,
                                               const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                               int accumulate, int nbm, int nbn, int *__restrict__ flags,
                                               float *__restrict__ parts, const int *__restrict__ order,
                                               const int *__restrict__ place, int *__restrict__ stats) {
  using S = Dma32Segment<BM, BN, KB, WM, WN, NBUF, true, EDGE, true, 0, MB>;
  using T = Dma32Tile<BM, BN, KB, WM, WN, NBUF, MB>;
  const int nk = (k + KB - 1) / KB;
  const int Tn = nbm * nbn, G = gridDim.x;
  const int xcd = blockIdx.x % NXCD, local = blockIdx.x / NXCD;
  const int gq = G / NXCD, gr = G % NXCD;
  const int rho = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + local;
  // (readfirstlane: what is loaded from memory or passed through LDS is workgroup-uniform, but hipcc cannot know -- and a
  // descriptor, LDS address or slice offset it takes for lane-dependent puts every LDS-DMA instruction of the loop into
  // a waterfall loop)
  const int q = __builtin_amdgcn_readfirstlane(order ? order[rho] : rho);
  const long long total = (long long)Tn * nk;
  const long long u0 = total * q / G, u1 = total * (q + 1) / G;
  if (u1 <= u0) return;
  const int t_first = (int)(u0 / nk), k_first = (int)(u0 % nk);
  const int t_last = (int)((u1 - 1) / nk), k_last_end = (int)(u1 - (long long)t_last * nk);
  auto tile_of = [&](int t, int &tm, int &tn) {   // grouped raster, no XCD remap (the ranges are XCD-contiguous)
    const int tt = __builtin_amdgcn_readfirstlane(place ? place[t] : t);
    const int per_group = GROUP_M * nbn;
    const int group = tt / per_group, first_m = group * GROUP_M;
    const int gsize = min(nbm - first_m, GROUP_M);
    const int in_group = tt - group * per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
  };
  // one lane's word made workgroup-uniform through the line of LDS behind the ring (the ring itself is never idle here)
  volatile int *word = reinterpret_cast<volatile int *>(lds + T::RING_BYTES / sizeof(float));
  auto uniform = [&](int v) {
    __syncthreads();
    if (threadIdx.x == 0) *word = v;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*word);
  };
  const bool whole_only = t_first == t_last && k_first == 0 && k_last_end == nk;
  const bool first_partial = !whole_only && k_first != 0, last_partial = !whole_only && k_last_end != nk;
  const int n_whole = t_last - t_first + 1 - (first_partial ? 1 : 0) - (last_partial ? 1 : 0);
  const int n_parts = (last_partial ? 1 : 0) + n_whole + (first_partial ? 1 : 0);
  enum { HEAD = 0, WHOLE = 1, TAIL = 2, LEFT_TO_US = 3 };
  struct Part { int t, kb, ke, kind; };
  auto part_at = [&](int s) {
    Part p;
    const int w = s - (last_partial ? 1 : 0);
    if (s == 0 && last_partial) p = Part{t_last, 0, k_last_end, HEAD};
    else if (w < n_whole) p = Part{t_first + (first_partial ? 1 : 0) + w, 0, nk, WHOLE};
    else p = Part{t_first, k_first, nk, TAIL};
    return p;
  };
  float *my_slot = parts + (size_t)q * BM * BN;
  typename S::Lane L;
  L.init(lda, ldb);
  typename S::Frag fr[2];
  Dma32Link link;
  int head_reply = 0;   // thread 0: what the word held when DONE went in
  if (last_partial && threadIdx.x == 0)   // "I am running": whoever needs the head may wait for it
    (void)__hip_atomic_fetch_or(&flags[t_last], SK_HEAD_RUNNING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int s = 0;; ++s) {
    Part p;
    if (s < n_parts) {
      p = part_at(s);
    } else {
      // after the range: a tail somebody left to us?  (head_reply is only looked at now: nobody stalls on an atomic's
      // round trip)
      if (s > n_parts || !last_partial || !(uniform(head_reply) & SK_TAIL_LEFT)) break;
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // our own write-through stores, read back through L2
        __hip_atomic_store(&flags[t_last], SK_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stats) __hip_atomic_fetch_add(stats, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      p = Part{t_last, k_last_end, nk, LEFT_TO_US};
    }
    Dma32Next nx;
    if (s + 1 < n_parts) {
      const Part f = part_at(s + 1);
      tile_of(f.t, nx.tm, nx.tn);
      nx.kb = f.kb;
      nx.len = f.ke - f.kb;
    }
    const float *part_in = nullptr;
    if (p.kind == TAIL) {
      int seen = SK_EMPTY;
      if (threadIdx.x == 0) {
        long long polls = 0;
        for (;;) {
          seen = __hip_atomic_load(&flags[t_first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (seen & SK_HEAD_DONE) break;
          if ((seen & SK_HEAD_RUNNING) && ++polls < (1ll << 22)) {   // resident and on its way: bounded by ITS OWN work
            __builtin_amdgcn_s_sleep(8);
            continue;
          }
          int expect = seen;                                        // not running (or the back-stop): leave the tail to it
          if (__hip_atomic_compare_exchange_strong(&flags[t_first], &expect, seen | SK_TAIL_LEFT, __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            seen = SK_TAIL_LEFT;
            break;
          }
        }
        if (seen & SK_HEAD_DONE) {
          seen = SK_HEAD_DONE;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          // the part that finishes a tile is the last reader of its word: it puts the 0 back
          __hip_atomic_store(&flags[t_first], SK_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (uniform(seen) != SK_HEAD_DONE) {
        // the head's owner is not running: it will find our mark and finish the tile itself.  The slices fetched
        // ahead for this tail are dropped -- once they have landed.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        link.primed = false;
        continue;
      }
      part_in = parts + (size_t)(q - 1) * BM * BN;
    } else if (p.kind == LEFT_TO_US) {
      part_in = my_slot;
    }
    int tm, tn;
    tile_of(p.t, tm, tn);
    S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, p.kb, p.ke, p.kb == 0 && accumulate != 0, part_in,
           p.kind == HEAD ? my_slot : nullptr, fr, link, nx);
    if (p.kind == HEAD) {
      // Publish (cdna guide G16, recipe R1): the partial tile went out write-through (sc1) -- every storing wave
      // drains its stores (the stream's loads in flight land with them), the workgroup meets, ONE lane ORs DONE in.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        head_reply = __hip_atomic_fetch_or(&flags[t_last], SK_HEAD_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int BM, int BN, int KB, int WM, int WN, int NBUF, bool EDGE = false, int MB = 1>
__global__ void __launch_bounds__(256)
sgemm_dma32_streamk_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B,
                           int ldb, float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn,
                           int *__restrict__ flags, float *__restrict__ parts, const int *__restrict__ order,
                           const int *__restrict__ place, int *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  streamk32_body<BM, BN, KB, WM, WN, NBUF, EDGE, MB>(lds, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nbm, nbn, flags, parts,
                                                  order, place, stats);
}

// the UNCHAINED persistent form (streamk_body over Dma32Seg: every part its own prologue) -- the A/B baseline
template <int BM, int BN, int KB, int WM, int WN, int NBUF, bool EDGE = false, int MB = 1>
__global__ void __launch_bounds__(256)
sgemm_dma32_streamk_unchained_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B,
                                     int ldb, float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn,
                                     int *__restrict__ flags, float *__restrict__ parts, const int *__restrict__ order,
                                     const int *__restrict__ place, int *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  streamk_body<Dma32Seg<BM, BN, KB, WM, WN, NBUF, EDGE, MB>>(lds, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nbm, nbn,
                                                          flags, parts, order, place, stats);
}

}  // namespace mmh
