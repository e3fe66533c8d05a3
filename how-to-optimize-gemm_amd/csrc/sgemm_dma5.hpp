// sgemm_dma5.hpp -- K2W: the LDS-DMA tiles with LOADER waves (round 4).
//
// Why it exists.  In K2L (sgemm_dma.hpp) each of a workgroup's four waves issues its share of the LDS-DMA pieces between
// its own MFMAs.  A `buffer_load_dwordx4 ... lds` keeps the issuing wave from issuing anything else for ~60-70 cycles
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"); the v_mfma_f32_16x16x4_f32 in front of it covers 32 of them, and
// with ONE workgroup per CU -- every size of the reference sweep below N = 1408 (cuda/parameters.h:5-7) and every
// persistent stream-K launch of the 128x128 tile (N = 2176 .. 2560) -- nobody else fills the rest: 8 pieces x ~38 idle
// cycles per 4096-cycle slice is the 7-8 % such launches sit below the many-workgroup sizes (profiles/r04_notes.md).
// Two rebuilds of the loop on 64-cycle matrix instructions (sgemm_dma32.hpp, tools build) hide the pieces and lose more
// elsewhere.  Here the pieces leave the MFMA waves altogether: NL extra waves do nothing but LDS-DMA -- the pieces of a
// K-slice dealt round-robin over them, NBUF - 1 slices ahead, one counted `s_waitcnt vmcnt` and the slice's barrier --
// and the four consumer waves run K2L's fragment reads and MFMAs and nothing else.  Same images, same fragments, same
// MFMA, same k order: the bits of every other kernel here.
//   * NL: one loader issues a 1 KiB piece per ~64 cycles = 16 B/clk; a 64x64 tile consumes exactly that (16 KiB per
//     1024 matrix-pipe cycles), so its loader is the bound -- it gets TWO; the larger tiles (24 KiB / 2048, 32 KiB /
//     4096 cycles) would be served by one, but one loader shares a SIMD with a consumer and holds it -- and at every
//     barrier the workgroup -- back: launch_dma5.hip instantiates FOUR for the 128-wide and the 96x64 tiles (measured,
//     profiles/r04_notes.md section 1).
//   * NBUF: ring depth.  A loader's `vmcnt` is a 6-bit counter, so (NBUF - 2) x (pieces per loader and slice) <= 63.
//   * D: fragments are read D k-steps ahead of the MFMAs that use them; with one wave per SIMD a ds_read_b32/_b64
//     stream reaches a fifth of the LDS rate (MI355X_MICROARCH.md, LDS: "from ~4 waves per SIMD"), so the one-
//     workgroup-per-CU launches want more reads in flight than the co-resident ones.
//   * RS (round 6): WHERE in the k-step those reads are issued.  Rounds 4-5 issued them as a block in front of the
//     k-step's MFMAs; a consumer wave alone on its SIMD feeds nothing to the matrix pipe while three to eight LDS
//     instructions leave (62 cycles per k-step on the 160x160 tile: the "slower loop" that kept it out of the product).
//     RS = 1: read unit j (an A pair, a B float or the B vector) follows MFMA j, in that instruction's 32 cycles of shadow.
//     Same slots, same fragments, same bits; +0.4 .. 5 % (profiles/r06_notes.md section 8).  RS = 0: the block form (tools build).
//
// Tiles of 32 i x 32 j (round 4: 96x96, 160x160, 160x96 -- the shapes that land N = 1536, 2560, 1920 of the reference
// sweep on one whole round of CUs).  The A side generalises at once (a fragment is one float per 16-row block).  For
// an odd number of 16-column blocks per wave (WTN = 3, 5) the B fragment is WTN single floats, lane li -> column
// 16 u + li of block u ("column-blocked"; the even widths keep K2L's WTN consecutive columns per lane and their vector
// C stores); a k-row of B is then 384 / 640 bytes, 64 lanes x 16 bytes do not divide it, and the piece -> (row, chunk)
// map repeats every 3 / 5 pieces = 8 k-rows: that many per-lane offsets, a scalar offset per group.  The two k-rows a
// 32-lane half of a ds_read_b32 touches (kq = 0, 1) are kept on different banks by XOR-ing bit 4 of the column with the
// row's parity -- on the source side of the DMA, as always.
//
// Thin edge tiles (EDGE instantiations).  One element past a tile boundary (N = 1025) pays a whole extra row and
// column of tiles; all but one 16-row (16-column) block of such a tile hold no valid element.  A wave with at most one
// valid block row (block column) runs a thin copy of the consumer side -- accumulators, a rolled K loop, stores -- that
// keeps block row 0 (block column 0) only (a block of rows >= m or columns >= n is never stored): the tile's fragment
// reads, barriers and DMA are unchanged, its matrix-pipe time drops to a half, a quarter or an eighth, and the whole
// tiles of the launch keep the unrolled, branch-free loop.  A plain launch dispatches such tiles last.
//
// The loader needs no per-piece address registers: a piece is eight consecutive A rows (or 256 / BN k-rows of B), so a
// lane's offset inside a piece is one VGPR per image and the piece's position is a scalar offset.
//
// Chained segments.  A persistent stream-K workgroup runs several (tile, K-range) segments back to back; K2L starts each
// with an empty pipeline.  The loaders instead walk ONE stream of slices through the ring: while the consumers finish
// a segment's last slices they already fetch the next segment's first, the fragment reads at the end of the
// last slice are the next segment's first, and the consumers' C / partial-tile stores -- and the drain that precedes a
// publish, which now waits for THEIR stores only -- run under loads in flight.  The ring position a segment starts at
// is then a run-time value: up to NBUF - 1 slices run from copies of the slice body in front of the unrolled ring loop.
#pragma once
#include <type_traits>

#include "ab_build.hpp"
#include "sgemm_dma.hpp"
#include "sgemm_mfma.hpp"   // the stream-K hand-over words (SK_*), GROUP_M

namespace mmh {

constexpr int dma5_gcd(int a, int b) { return b == 0 ? a : dma5_gcd(b, a % b); }

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, int NL = 1>
struct Dma5Tile {
  static_assert(KB == 32, "a K-slice row of A is 128 bytes: a piece of A is eight rows");
  static_assert(BM == 32 * WTM && BN == 32 * WTN, "four consumer waves as 2 x 2, wave tile 16 WTM x 16 WTN");
  static_assert(WTM >= 2 && WTM <= 6 && WTN >= 2 && WTN <= 6, "wave tile 32 .. 96 rows / columns");
  static_assert(NBUF >= 2 && NBUF <= 6, "ring depth (2: the next slice is requested after barrier kt - 1 and waited for before barrier kt -- K1W's two co-resident workgroups)");
  static_assert(NL == 1 || NL == 2 || NL == 4, "one, two or four loader waves");
  static constexpr int CONSUMERS = 4, LOADER = 4;   // wave index of the first loader
  static constexpr int WAVES_M = 2, WAVES_N = 2;
  static constexpr int THREADS = 64 * (CONSUMERS + NL);
  static constexpr bool BBLK = (WTN % 2) != 0;      // column-blocked B fragments (single floats)
  static constexpr int A_FLOATS = BM * KB, B_FLOATS = KB * BN, STAGE = A_FLOATS + B_FLOATS, KS = KB / 4;
  static constexpr int CHA = A_FLOATS / 256, CHB = B_FLOATS / 256, NP = CHA + CHB;   // 1 KiB pieces per slice
  static constexpr int NPL = NP / NL;                                                // ... per loader
  static constexpr int CPR_B = BN / 4;                                    // 16-byte chunks per k-row of B
  static constexpr int PB = CPR_B / dma5_gcd(64, CPR_B);                  // pieces until the lane -> (row, chunk) map repeats
  static constexpr int RB = 64 * PB / CPR_B;                              // k-rows those pieces hold
  static constexpr int LA = NBUF - 1;                                     // slices of look-ahead
  static_assert(CHA % NL == 0 && CHB % NL == 0, "the pieces of each image divide over the loaders");
  static_assert(CHB % PB == 0 && RB % 2 == 0, "whole periods; the row-parity swizzle must not depend on the period");
  static_assert((CHB / PB) % NL == 0, "B's piece periods divide over the loaders");
  static_assert((LA - 1) * NPL <= 63, "vmcnt is a 6-bit counter");
  // Raster group height: the tiles an XCD runs at one time (64 of the 128x64 tile: 32 CUs x 2) form a GM x 64 / GM patch
  // of the grouped raster; what its L2 must hold per K-slice is the patch's A rows + B columns.  Measured at N = 4096 on
  // the 128x64 tile (profiles/r04_notes.md, FETCH_SIZE x 2 per launch): GM = 1 / 2 / 4 / 8 / 16 / 32 -> 2.28 / 1.34 /
  // 1.07 / 1.34 / 2.29 / 4.35 GB -- 512 rows of A per patch is the sweet spot (the 64-row tiles' 8), and 4 beats 8 at
  // equal footprint: a k-row of B is contiguous, a K-slice of A one 128-byte line per row.  Time does not move.
  static constexpr int GM = 512 / BM >= 1 ? 512 / BM : 1;
  static constexpr size_t RING_BYTES = (size_t)NBUF * STAGE * sizeof(float);
  static constexpr size_t LDS_BYTES = RING_BYTES + 64;   // + the line the stream-K body passes a word through
  // the 16-byte chunk of B's k-row r that belongs at physical chunk position pc of the LDS row
  static __device__ __forceinline__ int src_chunk_b(int r, int pc) {
    if constexpr (BBLK) return pc ^ ((r & 1) << 2);          // columns +-16 on odd rows
    else if constexpr (WTN == 2) return pc ^ ((r & 1) << 3); // halves of a 64-float row swapped on odd rows (8-byte fragments)
    else return pc;
  }
};

// what follows a segment in its workgroup's stream, and the state carried from segment to segment
struct Dma5Next {
  int tm = 0, tn = 0, kb = 0, len = 0;
};
struct Dma5Link {
  int pos = 0;
  bool primed = false;
  bool fresh = true;   // nothing has run in this workgroup yet: the ring is idle, no wave holds a fragment read of it
};
// A stream-K HEAD part's publish, deferred into the part that follows it (a WHOLE tile: nothing it depends on): the
// partial tile went out as write-through stores; instead of draining them on the spot -- every consumer wave idle for
// a store round trip, then a workgroup barrier -- the drain rides on the next part's first slice barrier, by which
// time the stores have long completed, and one lane ORs DONE in after it.
// (run()'s pub_flag: the tile's hand-over word, null = nothing pending; pub_reply: thread 0's copy of what the word held
// when DONE went in.  Every slice's barrier point carries a wave-uniform test of "pending" -- two scalar instructions.)

// D: fragment prefetch distance in k-steps (ring of SLOTS register sets, slot = k-step mod SLOTS)
// What a RIM launch (sgemm_mfma_dma5_rim_kernel) hands to the segments of its tiles: the tiles cover the trimmed shape
// m0 x n0; the ONE row (r_m = 1) and / or ONE column (r_n = 1) beyond it are the rim.
struct Dma5Rim {
  int m0 = 0, n0 = 0, r_m = 0, r_n = 0;   // trimmed shape, rim rows / columns (0: none in that direction)
  bool do_row = false, do_col = false;    // this tile is in the last tile row / column of the trimmed grid
};

// (RIM: round 4's FUSED rim -- an extra wave per edge tile computing the N + 1 row / column on the vector ALU out of the
// tile's LDS; built, bit-exact, 2.2x slower per edge tile: the f32 MFMA and v_fma_f32 share the FMA lanes.  Its wave and
// its kernel live in tools/ab/sgemm_dma5_rim.hpp and exist in the tools build only; what stays here are the template
// parameter, the two extra DMA pieces of a RIM segment's loaders and this declaration.)
template <class Segment>
struct Dma5RimWave;

// VALU (round 6): the segment's consumer waves run K1W's vector-ALU loop (sgemm_valu_dma5.hpp: ds_read + v_pk_fma_f32, a
// thread tile of BM / 16 x BN / 16) instead of the MFMA one -- BASELINE.json configs[1]'s rung under the SAME loaders, ring
// protocol, chained segments and stream-K body.  What differs on this side of the barrier: A's image (K1W reads A along
// k: one bit of the row XORed into the chunk position, not three) and the fragment registers; D is then the A read's
// width in k-steps (2: ds_read_b64, 4: ds_read_b128).  Whole tiles only.
template <class Segment>
struct Dma5ValuConsumer;
template <int TI, int RJ, int AK>
struct Dma5ValuFrags {
  typedef float afrag_t __attribute__((ext_vector_type(AK)));
  afrag_t a[2][TI];   // the current and the next group of AK k-steps
  f32x4 b[4][RJ];     // k-step mod 4
};

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool PART_WT = false, bool EDGE = false, bool CHAIN = false,
          int NL = 1, int D = 2, bool RIM = false, bool VALU = false, int RS = 1>
struct Dma5Segment {
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>;
  static_assert(!VALU || (!EDGE && !RIM && (D == 2 || D == 4) && (BM == 64 || BM == 128) && (BN == 64 || BN == 128)),
                "the vector-ALU consumer: whole tiles, 16 x 16 threads of 4x4 output blocks");
  static constexpr bool kChain = CHAIN, kPartWt = PART_WT, kValu = VALU;
  static constexpr int kBM = BM, kBN = BN, kNBUF = NBUF, kAK = D;
  static_assert(!RIM || (BM == 64 && BN == 64 && WTN == 2 && NL == 2 && EDGE && !CHAIN), "the rim rides on the guarded 64x64 tile's plain launch");
  // RIM: floats behind the ring (and its word line): per ring position one 1 KiB piece of rim-row A values
  // [row e][32 k] and one of rim-column B values [k][4 columns]
  static constexpr int RIM_BASE = NBUF * T::STAGE + 16, RIM_STRIDE = 512;
  static constexpr size_t RIM_LDS_BYTES = T::LDS_BYTES + (RIM ? (size_t)NBUF * RIM_STRIDE * sizeof(float) : 0);
  static constexpr int NPL_R = T::NPL + (RIM ? 1 : 0);   // pieces per loader and slice
  static constexpr bool BBLK = T::BBLK;
  typedef float bfrag_t __attribute__((ext_vector_type(WTN)));
  typedef float afrag_t __attribute__((ext_vector_type(WTM)));
  typedef float c_vec_u __attribute__((ext_vector_type(WTN), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, bfrag_t>;
  static constexpr int SLOTS = D <= 3 ? 4 : 8;
  static_assert(D >= 1 && D < T::KS && T::KS % SLOTS == 0 && D < SLOTS, "fragment slots are numbered by k-step mod SLOTS");

  struct MfmaFrags {
    float a[SLOTS][WTM];
    float b[SLOTS][WTN];
  };
  using Frags = std::conditional_t<VALU, Dma5ValuFrags<BM / 16, BN / 64, D>, MfmaFrags>;

  // per-lane constants: the consumers' fragment addresses, the loaders' offsets inside a piece
  struct Lane {
    int wave, wm, wn, li, kq, ld;
    bool loader, rim;
    int stamp_base = 0;   // timeline build: the stream-K body moves it from part to part
    int a_off[8], b_off[BBLK ? WTN : 1];
    int v_tx = 0, v_wrow = 0, v_a_even = 0, v_a_odd = 0, v_b_col = 0, v_b_col_odd = 0;   // VALU consumers (sgemm_valu_dma5.hpp)
    uint32_t voff_a, voff_b[T::PB];
    __device__ __forceinline__ void init(int lda, int ldb) {
      const int tid = threadIdx.x, lane = tid & 63;
      wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      rim = RIM && wave == T::LOADER + NL;          // the rim wave (RIM launches only): behind the loaders
      loader = wave >= T::LOADER && !rim;
      ld = loader ? wave - T::LOADER : 0;
      wm = (wave & 3) / T::WAVES_N;
      wn = (wave & 3) % T::WAVES_N;
      li = lane & 15;
      kq = lane >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) a_off[j] = (wm * 16 * WTM + li) * KB + 4 * (j ^ (li & 7)) + kq;
      if constexpr (BBLK) {
#pragma unroll
        for (int u = 0; u < WTN; ++u) b_off[u] = T::A_FLOATS + kq * BN + 16 * ((wn * WTN + u) ^ (kq & 1)) + li;
      } else {
        b_off[0] = WTN == 4 ? T::A_FLOATS + kq * BN + wn * 64 + 4 * li
                            : T::A_FLOATS + kq * BN + 4 * ((wn * 8 + (li >> 1)) ^ ((kq & 1) << 3)) + 2 * (li & 1);
      }
      // loaders: the 16-byte chunk a lane fetches is the one that belongs at its (swizzled) position of the image
      {
        const int r = lane / 8, p = lane % 8;                              // piece j holds A rows 8 j + r
        voff_a = (uint32_t)(r * lda + 4 * (p ^ (VALU ? (r >> 1) & 1 : r & 7))) * 4u;
      }
#pragma unroll
      for (int jj = 0; jj < T::PB; ++jj) {
        const int c = 64 * jj + lane, r = c / T::CPR_B, pc = c % T::CPR_B; // piece PB g + jj holds k-rows RB g + r
        voff_b[jj] = (uint32_t)(r * ldb + 4 * T::src_chunk_b(r, pc)) * 4u;
      }
      if constexpr (VALU) {
        // a wave is 16 (tx) x 4 (t) threads: rows (BM / 4) wave + t + 4 i, columns 4 tx + 64 h (sgemm_valu_dma5.hpp)
        v_tx = lane & 15;
        v_wrow = (BM / 4) * (wave & 3) + (lane >> 4);
        const int a_bit = (v_wrow >> 1) & 1;
        v_a_even = v_wrow * KB + 4 * a_bit;
        v_a_odd = v_wrow * KB + 4 * (1 ^ a_bit);
        v_b_col = T::A_FLOATS + 4 * v_tx;
        v_b_col_odd = T::A_FLOATS + 4 * (BN == 64 ? (v_tx ^ 8) : v_tx);
      }
    }
  };

  static __device__ __forceinline__ void run(float *lds, const Lane &L, int m, int n, int k, const float *__restrict__ A,
                                             int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                             int tm, int tn, int kb, int ke, bool init_from_c, const float *part_in,
                                             float *part_out, Frags &fr, Dma5Link &link, const Dma5Next nx, int *pub_flag,
                                             int &pub_reply, const Dma5Rim rim = Dma5Rim{}) {
    constexpr int KS = T::KS, STAGE = T::STAGE, A_FLOATS = T::A_FLOATS, NPL = NPL_R, LA = T::LA;
    const int row0 = tm * BM, col0 = tn * BN;
    const int rows_valid = EDGE ? min(BM, m - row0) : BM;
    const int cols_valid = EDGE ? min(BN, n - col0) : BN;
    const bool chain = CHAIN && nx.len >= LA;
    int pos = CHAIN ? __builtin_amdgcn_readfirstlane(link.pos) : 0;
    const bool primed = CHAIN && __builtin_amdgcn_readfirstlane((int)link.primed) != 0;
    const bool ragged_k = EDGE && ke * KB > k;
    const int n_slices = ke - kb;
    if (CHAIN) {
      // where the stream stands after this segment
      link.pos = (pos + n_slices) % NBUF;
      link.primed = chain;
    }
    if (!primed) {
      // every wave is past its last fragment read of whatever ran before -- nothing did in front of a workgroup's FIRST
      // segment (round 5: that barrier made the loaders' first DMA wait for the consumers' per-lane setup)
      if constexpr (CHAIN) {
        if (!__builtin_amdgcn_readfirstlane((int)link.fresh)) __syncthreads();
      }
      pos = 0;
      if constexpr (CHAIN) link.pos = n_slices % NBUF;
    }
    if constexpr (CHAIN) link.fresh = false;

    if (L.loader) {
      // ------------------------------------------------------------------ the loader waves
      // descriptors as base + extent SCALARS, packed where they are used (a select between two 128-bit descriptors goes
      // through scratch memory)
      auto ext_a = [&](int valid) { return EDGE ? (uint32_t)(((valid - 1) * lda + k) * 4) : 0x7fffffffu; };
      auto ext_b = [&](int valid) { return EDGE ? (uint32_t)(((k - 1) * ldb + valid) * 4) : 0x7fffffffu; };
      const float *own_pa = A + (size_t)row0 * lda, *own_pb = B + col0;
      const uint32_t own_ea = ext_a(rows_valid), own_eb = ext_b(cols_valid);
      const float *next_pa = A, *next_pb = B;
      uint32_t next_ea = 0, next_eb = 0;   // no successor: the same instructions against empty descriptors
      if constexpr (CHAIN) {
        if (chain) {
          next_pa = A + (size_t)(nx.tm * BM) * lda;
          next_pb = B + nx.tn * BN;
          next_ea = ext_a(EDGE ? min(BM, m - nx.tm * BM) : BM);
          next_eb = ext_b(EDGE ? min(BN, n - nx.tn * BN) : BN);
        }
      }
      const int kdelta = nx.kb - ke;
      const int ld = L.ld;
      // RIM: loader 0 also fetches the rim row's A values of the slice (row m0: lanes 0-7 -> its eight 16-byte chunks;
      // everything past the descriptor's extent arrives as zeros), loader 1 the rim column's B values (a 4-byte piece:
      // lane -> B[k-row lane][n0], k contiguous in LDS); tiles off the rim run them against empty descriptors -- the
      // counted waits stay the same for every tile.
      uint32_t rim_ext = 0, rim_voff = 0;
      const float *rim_base = A;
      if constexpr (RIM) {
        const int lane = threadIdx.x & 63;
        if (ld == 0) {
          rim_base = A + (size_t)rim.m0 * lda;
          rim_ext = rim.do_row ? (uint32_t)(k * 4) : 0u;
          rim_voff = (uint32_t)((lane >> 3) * lda + 4 * (lane & 7)) * 4u;
        } else {
          rim_base = B + rim.n0;
          rim_ext = rim.do_col ? (uint32_t)(((k - 1) * ldb + 1) * 4) : 0u;
          rim_voff = (uint32_t)(lane * ldb) * 4u;
        }
      }
      auto issue = [&](float *buf, int kt, int rpos) {   // this loader's pieces of stream slice kt into ring buffer `buf` (position rpos)
        const bool own = kt < ke;
        const int ks = own ? kt : kt + kdelta;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(own ? own_pa : next_pa), 0,
                                                                            own ? own_ea : next_ea, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(own ? own_pb : next_pb), 0,
                                                                            own ? own_eb : next_eb, 0x00020000);
        const uint32_t off_a = (uint32_t)(ks * KB) * 4u, off_b = (uint32_t)(ks * KB) * (uint32_t)ldb * 4u;
        // pieces j = NL i + ld of each image (ld is wave-uniform: scalar arithmetic)
        static_for<T::CHA / NL>([&](auto i_c) {
          constexpr int i = decltype(i_c)::value;
          const int j = NL * i + ld;
          DmaPiece::one(ra, buf + 256 * j, L.voff_a, off_a + (uint32_t)(8 * j) * (uint32_t)lda * 4u);
        });
        // B: whole periods g = NL i2 + ld (PB pieces each, whose lane offsets are compile-time picks)
        static_for<T::CHB / T::PB / NL>([&](auto i_c) {
          constexpr int i2 = decltype(i_c)::value;
          const int g = NL * i2 + ld;
          static_for<T::PB>([&](auto jj_c) {
            constexpr int jj = decltype(jj_c)::value;
            DmaPiece::one(rb, buf + A_FLOATS + 256 * (T::PB * g + jj), L.voff_b[jj], off_b + (uint32_t)(T::RB * g) * (uint32_t)ldb * 4u);
          });
        });
        if constexpr (RIM) {
          const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rim_base), 0, own ? rim_ext : 0u, 0x00020000);
          float *rdst = lds + RIM_BASE + rpos * RIM_STRIDE + 256 * ld;
          if (ld == 0) DmaPiece::one(rr, rdst, rim_voff, off_a);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void *)rdst, 4, rim_voff, off_b, 0, 0);
        }
      };
      if (!primed) {
        static_for<LA>([&](auto s_c) {
          constexpr int S = decltype(s_c)::value;
          issue(lds + S * STAGE, kb + S, S);
          // (the first slice's pieces leave before the second slice's offsets are worked out: left alone hipcc computes
          // every piece's scalar offset of both slices in front of the first DMA -- ~40 instructions, round 5)
          __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NPL) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      int p2 = (pos + LA) % NBUF;   // ring position of stream slice kt + LA
      for (int kt = kb; kt < ke; ++kt) {
        issue(lds + p2 * STAGE, kt + LA, p2);   // into the buffer slice kt - 1 was read from (its barrier is behind us)
        p2 = p2 == NBUF - 1 ? 0 : p2 + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NPL) : "memory");   // stream slice kt + 1 is whole
        __builtin_amdgcn_s_barrier();
      }
      if (!chain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may still be landing in LDS
      return;
    }

    if constexpr (RIM) {   // (tools build: the fused rim's wave, tools/ab/sgemm_dma5_rim.hpp)
      if (L.rim) {
        Dma5RimWave<Dma5Segment>::run(lds, m, n, k, C, ldc, row0, col0, kb, ke, init_from_c, rim);
        return;
      }
    }

    // -------------------------------------------------------------------- the consumer waves
    if constexpr (VALU) {
      Dma5ValuConsumer<Dma5Segment>::consume(lds, L, C, ldc, row0, col0, kb, ke, pos, primed, chain, init_from_c, part_in, part_out, fr,
                                             pub_flag, pub_reply);
    } else {
    // EDGE: how many of this wave's 16-row / 16-column blocks hold a valid element (wave-uniform).  Block t holds rows
    // 16 (WTM wm + t) .. + 15 of the tile; block u holds, column-blocked, columns 16 (WTN wn + u) .. + 15, and with
    // WTN consecutive columns per lane the columns 16 WTN wn + WTN li + u -- its first (li = 0) is its smallest.
    // A wave with at most ONE valid block row (or block column) runs a thin copy of everything below that keeps,
    // computes and stores block row 0 (block column 0) only: what a shape a few elements past a tile boundary needs
    // (N = 1025: an edge tile's matrix-pipe time drops to a half, a quarter or -- the waves with no valid block --
    // an MFMA per k-step).  The copies are whole (accumulators, K loop, stores): no accumulator ever merges from two
    // paths, so the whole tiles' registers and loop are what they are without the thin forms.
    if constexpr (EDGE) {
      const int rv = rows_valid - L.wm * 16 * WTM, cv = cols_valid - L.wn * 16 * WTN;
      const bool thin_m = __builtin_amdgcn_readfirstlane((int)(rv <= 16)) != 0;
      const bool thin_n = __builtin_amdgcn_readfirstlane((int)(BBLK ? cv <= 16 : cv <= 1)) != 0;
      if (thin_m && thin_n) {
        consume(std::true_type{}, std::true_type{}, lds, L, m, n, k, C, ldc, row0, col0, rows_valid, cols_valid, kb, ke, pos, primed, chain,
                init_from_c, part_in, part_out, fr, pub_flag, pub_reply);
        return;
      }
      if (thin_m) {
        consume(std::true_type{}, std::false_type{}, lds, L, m, n, k, C, ldc, row0, col0, rows_valid, cols_valid, kb, ke, pos, primed, chain,
                init_from_c, part_in, part_out, fr, pub_flag, pub_reply);
        return;
      }
      if (thin_n) {
        consume(std::false_type{}, std::true_type{}, lds, L, m, n, k, C, ldc, row0, col0, rows_valid, cols_valid, kb, ke, pos, primed, chain,
                init_from_c, part_in, part_out, fr, pub_flag, pub_reply);
        return;
      }
    }
    consume(std::false_type{}, std::false_type{}, lds, L, m, n, k, C, ldc, row0, col0, rows_valid, cols_valid, kb, ke, pos, primed, chain,
            init_from_c, part_in, part_out, fr, pub_flag, pub_reply);
  }   // (!VALU)
  }

  // The consumer side of a segment with NT x NU of the wave's WTM x WTN blocks kept (all of them, or -- thin edge
  // tiles -- block row 0 / block column 0 only).
  template <class TM1, class TN1>
  static __device__ __forceinline__ void consume(TM1, TN1, float *lds, const Lane &L, int m, int n, int k, float *__restrict__ C,
                                                 int ldc, int row0, int col0, int rows_valid, int cols_valid, int kb, int ke,
                                                 int pos, bool primed, bool chain, bool init_from_c, const float *part_in,
                                                 float *part_out, Frags &fr, int *pub_flag, int &pub_reply) {
    constexpr int KS = T::KS, STAGE = T::STAGE;
    constexpr bool THIN = TM1::value || TN1::value;
    constexpr int NT = TM1::value ? 1 : WTM, NU = TN1::value ? 1 : WTN;   // blocks kept
    const bool ragged_k = EDGE && ke * KB > k;
    // a thin wave's K loop is a chain of short steps (a fragment read, one or two MFMAs): it runs at the front of its
    // SIMD's issue order, beside the co-resident whole tile's waves whose MFMAs fill the matrix pipe either way
    if constexpr (THIN) __builtin_amdgcn_s_setprio(3);
    const int crow = row0 + L.wm * 16 * WTM + 4 * L.kq;
    // even widths: this lane's WTN consecutive columns; column-blocked: column of block 0 (block u: + 16 u)
    const int ccol = BBLK ? col0 + L.wn * 16 * WTN + L.li : col0 + L.wn * 16 * WTN + WTN * L.li;
    const bool whole_c = !EDGE || (!THIN && rows_valid == BM && cols_valid == BN);
    constexpr bool VEC = !BBLK && !THIN;   // this lane's columns of a row are one vector
    auto col_of = [&](int u) { return BBLK ? ccol + 16 * u : ccol + u; };
    f32x4 acc[NT][NU];
    if (part_in) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float *src = part_in + (size_t)(crow + 16 * t + r - row0) * BN + (ccol - col0);
          if constexpr (VEC) {
            const bfrag_t v = *reinterpret_cast<const bfrag_t *>(src);
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[t][u][r] = v[u];
          } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[t][u][r] = src[BBLK ? 16 * u : u];
          }
        }
    } else if (init_from_c) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = crow + 16 * t + r;
          float v[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) v[u] = 0.0f;
          if constexpr (VEC) {
            if (whole_c) {
              const bfrag_t w = *reinterpret_cast<const c_vec *>(C + (size_t)row * ldc + ccol);
#pragma unroll
              for (int u = 0; u < NU; ++u) v[u] = w[u];
            } else if (row < m) {
#pragma unroll
              for (int u = 0; u < NU; ++u)
                if (col_of(u) < n) v[u] = C[(size_t)row * ldc + col_of(u)];
            }
          } else if (whole_c || row < m) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
              if (whole_c || col_of(u) < n) v[u] = C[(size_t)row * ldc + col_of(u)];
          }
#pragma unroll
          for (int u = 0; u < NU; ++u) acc[t][u][r] = v[u];
        }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto frag_a = [&](const float *buf, auto ks_c, float (&a)[WTM]) {
      constexpr int ks = decltype(ks_c)::value;
#pragma unroll
      for (int t = 0; t < WTM; ++t) a[t] = buf[L.a_off[ks & 7] + 4 * (ks & ~7) + t * 16 * KB];
    };
    auto frag_b = [&](const float *buf, auto ks_c, float (&b)[WTN]) {
      constexpr int ks = decltype(ks_c)::value;
      if constexpr (BBLK) {
#pragma unroll
        for (int u = 0; u < WTN; ++u) b[u] = buf[L.b_off[u] + 4 * ks * BN];
      } else {
        const bfrag_t v = *reinterpret_cast<const bfrag_t *>(buf + L.b_off[0] + 4 * ks * BN);
#pragma unroll
        for (int u = 0; u < WTN; ++u) b[u] = v[u];
      }
    };
    if (!primed) {
      __builtin_amdgcn_s_barrier();   // the loaders have the first slice in LDS
      static_for<D>([&](auto d_c) {
        constexpr int d = decltype(d_c)::value;
        frag_a(lds, d_c, fr.a[d]);
        frag_b(lds, d_c, fr.b[d]);
      });
    }
    dma_stamp(L.stamp_base + 1);

    // One K-slice out of ring buffer `buf` (K2L's slice body without its DMA pieces): per k-step the fragment reads
    // for k-step ks + D (all of them, thin or not: the next segment of a chain may be a whole tile), then the MFMAs of
    // k-step ks; before k-step KS - D the slice's barrier -- from there on the reads go to the NEXT buffer (the loaders'
    // counted wait says it is whole), and every read of this one has been issued.
    // a HEAD's deferred publish (see Dma5Link's neighbour above): pending until the first slice barrier of this part
    bool pend = CHAIN && __builtin_amdgcn_readfirstlane((int)(pub_flag != nullptr)) != 0;
    auto slice_at = [&](int kt, const float *buf, const float *nxt, auto tail_c) {
      constexpr bool TAIL = decltype(tail_c)::value;   // EDGE: a slice that may hold k's past the end (operands masked)
      const int krem = TAIL ? k - kt * KB : KB;
      static_for<KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks == KS - D) {
          if constexpr (CHAIN) {
            if (pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial-tile stores have completed
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if constexpr (CHAIN) {
            if (pend) {   // ... and so have every other consumer wave's: ONE lane sets DONE
              if (threadIdx.x == 0)
                pub_reply = __hip_atomic_fetch_or(pub_flag, SK_HEAD_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              pend = false;
            }
          }
        }
        // RS (read spread): the fragment reads of k-step ks + D leave ONE AT A TIME behind the first MFMAs of k-step ks,
        // each in the shadow of a matrix instruction, instead of as a block in front of them (a wave alone on its SIMD
        // issues nothing else while eight ds_read instructions leave: the 160x160 tile's 5 + 5 single-float fragments)
        constexpr int UA = (WTM + 1) / 2, UB = BBLK ? WTN : 1, UNITS = UA + UB;
        auto read_unit = [&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          constexpr int rks = ks + D < KS ? ks + D : ks + D - KS;
          const float *rb = ks + D < KS ? buf : nxt;
          float(&fa)[WTM] = fr.a[(ks + D) % SLOTS];
          float(&fb)[WTN] = fr.b[(ks + D) % SLOTS];
          if constexpr (j < UA) {
#pragma unroll
            for (int t = 2 * j; t < 2 * j + 2 && t < WTM; ++t) fa[t] = rb[L.a_off[rks & 7] + 4 * (rks & ~7) + t * 16 * KB];
          } else if constexpr (BBLK) {
            fb[j - UA] = rb[L.b_off[j - UA] + 4 * rks * BN];
          } else {
            frag_b(rb, std::integral_constant<int, rks>{}, fb);
          }
        };
        if constexpr (RS == 0) {
        if constexpr (ks + D < KS) {
          frag_a(buf, std::integral_constant<int, ks + D>{}, fr.a[(ks + D) % SLOTS]);
          frag_b(buf, std::integral_constant<int, ks + D>{}, fr.b[(ks + D) % SLOTS]);
        } else {
          frag_a(nxt, std::integral_constant<int, ks + D - KS>{}, fr.a[(ks + D) % SLOTS]);
          frag_b(nxt, std::integral_constant<int, ks + D - KS>{}, fr.b[(ks + D) % SLOTS]);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        float a[NT], b[NU];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = fr.a[ks % SLOTS][t];
#pragma unroll
        for (int u = 0; u < NU; ++u) b[u] = fr.b[ks % SLOTS][u];
        if constexpr (TAIL) {
          // A's columns past k are the next row's floats or the caller's padding (NaN included): zero this lane's
          // operands of the k's that do not exist (B's rows there are zeros by descriptor; belt and braces)
          const bool live = 4 * ks + L.kq < krem;
#pragma unroll
          for (int t = 0; t < NT; ++t) a[t] = live ? a[t] : 0.0f;
#pragma unroll
          for (int u = 0; u < NU; ++u) b[u] = live ? b[u] : 0.0f;
        }
        if constexpr (RS == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int u = 0; u < NU; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
        } else {
          static_for<NT * NU>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value, t = i / NU, u = i % NU;
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
            if constexpr (i < UNITS) {
              __builtin_amdgcn_sched_barrier(0);
              read_unit(i_c);
              __builtin_amdgcn_sched_barrier(0);
            }
          });
          static_for<UNITS>([&](auto j_c) {   // (thin waves: fewer MFMAs than reads)
            if constexpr (decltype(j_c)::value >= NT * NU) read_unit(j_c);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    auto slice = [&](int kt, auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % NBUF;
      slice_at(kt, lds + CUR * STAGE, lds + NXT * STAGE, std::false_type{});
    };
    const int ke_main = ragged_k ? ke - 1 : ke;
    int kt = kb;
    {
      // (thin waves run the same unrolled ring: a rolled copy with a run-time ring position was measured first -- hipcc
      // reuses the address registers as fragment destinations there and waits for EVERY read before the next MFMA, one
      // LDS round trip per k-step: 1.05 us per slice, twice a whole tile's, profiles/r04_notes.md)
      // ONE exit per loop (with `break`s between the unrolled slices hipcc copies the accumulators on the hot path)
      if constexpr (CHAIN) {   // up to NBUF - 1 slices to reach ring position 0
        static_for<NBUF - 1>([&](auto p_c) {
          constexpr int P = decltype(p_c)::value + 1;
          if (pos == P && kt < ke_main) {
            slice(kt, std::integral_constant<int, P>{});
            ++kt;
            pos = (P + 1) % NBUF;
          }
        });
      }
      while (kt + NBUF <= ke_main) {
        static_for<NBUF>([&](auto c_c) { slice(kt + decltype(c_c)::value, c_c); });
        kt += NBUF;
      }
      static_for<NBUF - 1>([&](auto c_c) {   // (only reached at ring position 0)
        constexpr int CUR = decltype(c_c)::value;
        if (kt < ke_main) {
          slice(kt, std::integral_constant<int, CUR>{});
          ++kt;
          pos = CUR + 1;
        }
      });
      if constexpr (EDGE) {
        if (ragged_k) {   // one slice per tile pays for a run-time ring position (an address add per fragment read)
          const int nx1 = pos == NBUF - 1 ? 0 : pos + 1;
          slice_at(kt, lds + pos * STAGE, lds + nx1 * STAGE, std::true_type{});
        }
      }
    }
    if (!chain) {
      // keep the fragments prefetched past the last slice formally alive (see sgemm_dma.hpp)
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
#pragma unroll
        for (int u = 0; u < WTN; ++u) asm volatile("" ::"v"(fr.b[i][u]));
#pragma unroll
        for (int t = 0; t < WTM; ++t) asm volatile("" ::"v"(fr.a[i][t]));
      }
    }
    dma_stamp(L.stamp_base + 2);
    if constexpr (THIN) __builtin_amdgcn_s_setprio(0);

    auto out_vec = [&](int t, int r) {
      bfrag_t v;
#pragma unroll
      for (int u = 0; u < WTN; ++u) v[u] = u < NU ? acc[t][u < NU ? u : 0][r] : 0.0f;
      return v;
    };
    if (part_out) {
      __amdgpu_buffer_rsrc_t rsrc_p;
      if constexpr (PART_WT) rsrc_p = __builtin_amdgcn_make_buffer_rsrc(part_out, 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = crow + 16 * t + r;
          const bfrag_t v = out_vec(t, r);
          const uint32_t off = (uint32_t)(((row - row0) * BN + (ccol - col0)) * 4);
          if constexpr (!VEC) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
              constexpr int STEP = BBLK ? 16 : 1;
              if constexpr (PART_WT) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)v[u]), rsrc_p, off + 4u * STEP * u, 0, 16);
              else part_out[(size_t)(row - row0) * BN + (ccol - col0) + STEP * u] = v[u];
            }
          } else if constexpr (PART_WT) {
            if constexpr (WTN == 4) {
              typedef int i32x4_t __attribute__((ext_vector_type(4)));
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsrc_p, off, 0, 16);
            } else {
              static_assert(!VEC || WTN == 2 || WTN == 4 || !PART_WT, "write-through partial tiles: 8- or 16-byte vectors");
              typedef int i32x2_t __attribute__((ext_vector_type(2)));
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, v), rsrc_p, off, 0, 16);
            }
          } else {
            *reinterpret_cast<bfrag_t *>(part_out + (size_t)(row - row0) * BN + (ccol - col0)) = v;
          }
        }
    } else if (VEC && whole_c) {
      if constexpr (VEC) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<c_vec *>(C + (size_t)(crow + 16 * t + r) * ldc + ccol) = out_vec(t, r);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = crow + 16 * t + r;
          const bfrag_t v = out_vec(t, r);
          if (whole_c || row < m) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
              if (whole_c || col_of(u) < n) C[(size_t)row * ldc + col_of(u)] = v[u];
          }
        }
    }
  }
};

// One workgroup per C tile (XCD-aware block -> tile map), whole K range.
// EDGE: a last tile row / column that is THIN (at most 16 valid rows / columns: its waves run the thin copies of the
// consumer) is taken out of the raster and dispatched LAST: the whole tiles fill the CUs first, in the order a launch
// of the trimmed shape would, and the thin tiles -- a fraction of a whole tile's matrix-pipe time each -- land beside
// them as second workgroups.  In raster order they would take first-round slots and push whole tiles into a second
// round (N = 1025: 289 tiles of 64x64 for 256 CUs, 33 of them thin).
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE = false, int NL = 1, int D = 2, int RS = 1>
__global__ void __launch_bounds__(64 * (4 + NL))
sgemm_mfma_dma5_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                       float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = Dma5Segment<BM, BN, KB, WTM, WTN, NBUF, false, EDGE, false, NL, D, false, false, RS>;
  int tm, tn;
  // (round 5: every kernel argument is requested HERE -- left alone hipcc loads the operands' pointers and leading
  // dimensions only behind the branch to the loader path, a second scalar-memory round trip in front of the first DMA)
  asm volatile("" ::"s"(A), "s"(B), "s"(C), "s"(lda), "s"(ldb), "s"(ldc), "s"(k));
  dma_stamp(0);
  // TAIL SPLIT (bits 16-31 of `accumulate`: the id of this launch's first workgroup, / 8; launch_dma5.hip decides).  The
  // dispatcher hands a launch's first workgroups out one per CU -- ids 0 .. CUs - 1 --, then the second slot of every CU, then
  // the third; they start together and, the kernels being what they are, the ones that share a CU END together: all slots of
  // a CU are refilled at once, and a last round of fewer tiles than CUs lands two (three) per CU on a part of the chip instead
  // of one per CU -- a whole extra round (4822 x 1268 x 2551, 760 tiles of 128x64: 110 TFLOP/s, 143 with one per CU).  Such a
  // launch goes out as TWO: the whole rounds, then the last round as a launch of its own -- whose workgroups are "first
  // workgroups" again, one per CU -- continuing the first one's ids.  (Round 6 first proved the diagnosis with a stagger of the
  // first round's second slots: +20-29 % where it held, but a tile-dependent delay and -14 % elsewhere;
  // profiles/r06_first_round_stagger_ab.md.)
  const unsigned bid = blockIdx.x + (((unsigned)accumulate >> 16) << 3);
  accumulate &= 0xffff;
  if constexpr (EDGE) {
    const int thin_row = (nbm > 1 && m - (nbm - 1) * BM <= 16) ? 1 : 0, thin_col = (nbn > 1 && n - (nbn - 1) * BN <= 16) ? 1 : 0;
    const int nbm_f = nbm - thin_row, nbn_f = nbn - thin_col, n_full = nbm_f * nbn_f;
    int r = (int)bid - n_full;
    if (r < 0) {
      block_to_tile_g(bid, n_full, nbm_f, nbn_f, Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>::GM, tm, tn);
    } else if (thin_col && r < nbm) {   // the thin column, top to bottom (its corner with a thin row included)
      tm = r;
      tn = nbn - 1;
    } else {                            // the thin row, left to right
      if (thin_col) r -= nbm;
      tm = nbm - 1;
      tn = r;
    }
  } else {
    int gm = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>::GM;
    if constexpr (kAbBuild) {   // tools build: the raster group height rides in bits 8-15 of `accumulate` (0: the tile's own)
      const int ab_gm = (accumulate >> 8) & 0xff;
      accumulate &= 1;
      if (ab_gm > 0) gm = ab_gm;
    }
    block_to_tile_g(bid, nbm * nbn, nbm, nbn, gm, tm, tn);
  }
  typename S::Lane L;
  L.init(lda, ldb);
  typename S::Frags fr;
  Dma5Link link;
  int no_reply = 0;
  S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, 0, (k + KB - 1) / KB, accumulate != 0, nullptr, nullptr, fr, link,
         Dma5Next{}, nullptr, no_reply);
  dma_stamp_after_stores(3);
}


// ---------------------------------------------------------------------------------------------------------------
// K2Wp: the chained stream-K body.  Ranges, the order of a range's parts (head of the last tile FIRST, whole tiles, tail
// of the first tile LAST), the hand-over protocol and its words are streamk_body's (sgemm_mfma.hpp, K2p); the parts run
// as ONE stream of K-slices (Dma5Segment, CHAIN).  A tail's first slices are fetched BEFORE its hand-over word is
// looked at (they depend on nobody); should the word say the head's owner is not running (the wait-free path: leave),
// they are dropped.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE, bool CHAINED, int NL, int D, bool VALU = false, int RS = 1>
__device__ __forceinline__ void streamk5_body(float *lds, int m, int n, int k, const float *__restrict__ A, int lda,
                                              const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                              int accumulate, int nbm, int nbn, int *__restrict__ flags,
                                              float *__restrict__ parts, const int *__restrict__ order,
                                              const int *__restrict__ place, int *__restrict__ stats) {
  constexpr bool chained = CHAINED;
  using S = Dma5Segment<BM, BN, KB, WTM, WTN, NBUF, true, EDGE, true, NL, D, false, VALU, RS>;
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>;
  // tools build: bit 1 of `accumulate` = publish every head on the spot, bit 2 = whole-tile ranges, bits 8-15 = raster group height
  const bool ab_nodefer = kAbBuild && (accumulate & 2) != 0;
  const bool ab_whole = kAbBuild && (accumulate & 4) != 0;
  const int ab_gm = kAbBuild ? (accumulate >> 8) & 0xff : 0;
  if constexpr (kAbBuild) accumulate &= 1;
  const int GMr = ab_gm > 0 ? ab_gm : T::GM;
  const int nk = (k + KB - 1) / KB;
  const int Tn = nbm * nbn, G = gridDim.x;
  const int xcd = blockIdx.x % NXCD, local = blockIdx.x / NXCD;
  const int gq = G / NXCD, gr = G % NXCD;
  const int rho = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + local;
  // (readfirstlane: what is loaded from memory or passed through LDS is workgroup-uniform, but hipcc cannot know -- and a
  // descriptor, LDS address or slice offset it takes for lane-dependent puts every LDS-DMA instruction into a waterfall loop)
  const int q = __builtin_amdgcn_readfirstlane(order ? order[rho] : rho);
  // range q of G over Tn x nk slices: [total q / G, total (q + 1) / G) -- in 32-bit arithmetic (the launcher keeps
  // total < 2^31: launch_streamk): total q / G = (total / G) q + ((total % G) q) / G, and (total % G) q < G^2.  (The
  // 64-bit divisions this replaces were a microsecond of scalar code in front of every workgroup's first DMA.)
  const unsigned total = (unsigned)Tn * (unsigned)nk;
  unsigned per, rem, u0, u1;
  if ((G & (G - 1)) == 0) {   // (256 / 512 persistent workgroups: four divisions, ~110 scalar instructions, become shifts -- round 5)
    const int sh = __builtin_ctz((unsigned)G);
    per = total >> sh;
    rem = total & (unsigned)(G - 1);
    u0 = per * (unsigned)q + ((rem * (unsigned)q) >> sh);
    u1 = per * (unsigned)(q + 1) + ((rem * (unsigned)(q + 1)) >> sh);
  } else {
    per = total / (unsigned)G;
    rem = total % (unsigned)G;
    u0 = per * (unsigned)q + rem * (unsigned)q / (unsigned)G;
    u1 = per * (unsigned)(q + 1) + rem * (unsigned)(q + 1) / (unsigned)G;
  }
  if constexpr (kAbBuild) {
    if (ab_whole) {   // ranges rounded to tile boundaries: range q takes tiles [Tn q / G, Tn (q + 1) / G)
      u0 = (unsigned)(((unsigned long long)Tn * (unsigned)q) / (unsigned)G) * (unsigned)nk;
      u1 = (unsigned)(((unsigned long long)Tn * (unsigned)(q + 1)) / (unsigned)G) * (unsigned)nk;
    }
  }
  if (u1 <= u0) return;
  const int t_first = (int)(u0 / (unsigned)nk), k_first = (int)(u0 - (unsigned)t_first * (unsigned)nk);
  const int t_last = (int)((u1 - 1) / (unsigned)nk), k_last_end = (int)(u1 - (unsigned)t_last * (unsigned)nk);
  auto tile_of = [&](int t, int &tm, int &tn) {   // grouped raster, no XCD remap (the ranges are XCD-contiguous)
    const int tt = __builtin_amdgcn_readfirstlane(place ? place[t] : t);
    const int per_group = GMr * nbn;
    const int group = tt / per_group, first_m = group * GMr;
    const int gsize = min(nbm - first_m, GMr);
    const int in_group = tt - group * per_group;
    if ((gsize & (gsize - 1)) == 0) {
      tm = first_m + (in_group & (gsize - 1));
      tn = in_group >> __builtin_ctz((unsigned)gsize);
    } else {
      tn = in_group / gsize;
      tm = first_m + in_group - tn * gsize;
    }
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
#ifdef MMH_DMA_TIMELINE   // tools/sk_timeline.py: slot 0 entry, 1 exit, 2 part count | kinds, 3 part lengths, 4 + 4 s ... part s
  dma_stamp(0);
  {
    unsigned long long kinds = (unsigned long long)n_parts, lens = 0;
    for (int s = 0; s < n_parts && s < 7; ++s) {
      const Part p = part_at(s);
      kinds |= (unsigned long long)p.kind << (8 + 2 * s);
      lens |= (unsigned long long)min(p.ke - p.kb, 255) << (8 * s);
    }
    dma_stamp_value(2, kinds);
    dma_stamp_value(3, lens);
  }
#endif
  typename S::Lane L;
  L.init(lda, ldb);
  typename S::Frags fr;
  Dma5Link link;
  int head_reply = 0;   // thread 0: what the word held when DONE went in
  if (last_partial && threadIdx.x == 0)   // "I am running": whoever needs the head may wait for it
    (void)__hip_atomic_fetch_or(&flags[t_last], SK_HEAD_RUNNING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Publish (cdna guide G16, recipe R1): the partial tile went out write-through (sc1) -- every storing wave drains ITS
  // stores (the loaders' loads in flight are their own business), the workgroup meets, ONE lane ORs DONE in.
  auto publish_now = [&]() {
    if (!L.loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      head_reply = __hip_atomic_fetch_or(&flags[t_last], SK_HEAD_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  int *pending = nullptr;       // a HEAD's hand-over word whose DONE rides on the next part's first slice (chained parts only)
  int early = SK_EMPTY;         // thread 0: the TAIL's hand-over word as read BEFORE the head's store drain
  bool have_early = false;
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
    Dma5Next nx;
    int next_kind = -1;
    if (s + 1 < n_parts) {
      const Part f = part_at(s + 1);
      next_kind = f.kind;
      if (chained) {
        tile_of(f.t, nx.tm, nx.tn);
        nx.kb = f.kb;
        nx.len = f.ke - f.kb;
      }
    }
    const float *part_in = nullptr;
#ifdef MMH_DMA_TIMELINE
    L.stamp_base = 4 + 4 * min(s, 6);
    dma_stamp(L.stamp_base);
#endif
    if (p.kind == TAIL) {
      int seen = SK_EMPTY;
      if (threadIdx.x == 0) {
        long long polls = 0;
        for (;;) {
          if (have_early) {
            seen = early;   // (read while our own head's stores were draining: no round trip of its own)
            have_early = false;
          } else {
            seen = __hip_atomic_load(&flags[t_first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
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
        // ahead for this tail are dropped -- once they have landed (the loaders' wait).
        if (L.loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        link.primed = false;
        continue;
      }
      part_in = parts + (size_t)(q - 1) * BM * BN;
    } else if (p.kind == LEFT_TO_US) {
      part_in = my_slot;
    }
    int tm, tn;
    tile_of(p.t, tm, tn);
    tm = __builtin_amdgcn_readfirstlane(tm);
    tn = __builtin_amdgcn_readfirstlane(tn);
    const int pkb = __builtin_amdgcn_readfirstlane(p.kb), pke = __builtin_amdgcn_readfirstlane(p.ke);
    nx.tm = __builtin_amdgcn_readfirstlane(nx.tm);
    nx.tn = __builtin_amdgcn_readfirstlane(nx.tn);
    nx.kb = __builtin_amdgcn_readfirstlane(nx.kb);
    nx.len = __builtin_amdgcn_readfirstlane(nx.len);
    int *const pub_flag = pending;
    pending = nullptr;
    S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, pkb, pke, pkb == 0 && accumulate != 0, part_in,
           p.kind == HEAD ? my_slot : nullptr, fr, link, nx, pub_flag, head_reply);
    if (p.kind == HEAD) {
      if (chained && !ab_nodefer && next_kind == WHOLE && nk >= 3) {
        // a whole tile follows, which depends on nobody: the publish rides on its first slice's barrier (Dma5Publish)
        pending = &flags[t_last];
      } else {
        if (next_kind == TAIL && threadIdx.x == 0) {
          // our tail's hand-over word, read while our head's stores drain (one round trip instead of two)
          early = __hip_atomic_load(&flags[t_first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          have_early = true;
        }
        publish_now();
      }
    }
#ifdef MMH_DMA_TIMELINE
    dma_stamp(L.stamp_base + 3);
#endif
  }
#ifdef MMH_DMA_TIMELINE
  dma_stamp_after_stores(1);
#endif
}

// CHAINED = false: every part of a range starts with an empty pipeline (the A/B baseline, tools build)
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE = false, bool CHAINED = true, int NL = 1, int D = 2, int RS = 1>
__global__ void __launch_bounds__(64 * (4 + NL))
sgemm_dma5_streamk_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B,
                          int ldb, float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn,
                          int *__restrict__ flags, float *__restrict__ parts, const int *__restrict__ order,
                          const int *__restrict__ place, int *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  asm volatile("" ::"s"(A), "s"(B), "s"(C), "s"(lda), "s"(ldb), "s"(ldc), "s"(k), "s"(flags), "s"(parts), "s"(order), "s"(place));   // (every argument requested at entry: sgemm_mfma_dma5_kernel)
  streamk5_body<BM, BN, KB, WTM, WTN, NBUF, EDGE, CHAINED, NL, D, false, RS>(lds, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nbm, nbn,
                                                                   flags, parts, order, place, stats);
}

}  // namespace mmh
