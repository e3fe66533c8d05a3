// sgemm_dma5.hpp -- K2W: the LDS-DMA tiles with a LOADER wave (round 4).
//
// Why it exists.  In K2L (sgemm_dma.hpp) each of a workgroup's four waves issues its share of the LDS-DMA pieces between
// its own MFMAs.  A `buffer_load_dwordx4 ... lds` keeps the issuing wave from issuing anything else for ~60-70 cycles
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"); the v_mfma_f32_16x16x4_f32 in front of it covers 32 of them, and
// with ONE workgroup per CU -- every size of the reference sweep below N = 1408 (cuda/parameters.h:5-7) and every
// persistent stream-K launch of the 128x128 tile (N = 2176 .. 2560) -- nobody else fills the rest: 8 pieces x ~38 idle
// cycles per 4096-cycle slice is the 7-8 % such launches sit below the many-workgroup sizes (profiles/r04_notes.md).
// Two rebuilds of the loop on 64-cycle matrix instructions (sgemm_dma32.hpp, tools build) hide the pieces and lose more
// elsewhere.  Here the pieces leave the MFMA waves altogether: a FIFTH wave does nothing but LDS-DMA -- all pieces of a
// K-slice, two slices ahead, one counted `s_waitcnt vmcnt` and the slice's barrier -- and the four consumer waves run
// K2L's fragment reads and MFMAs and nothing else.  Same images, same fragments, same MFMA, same k order: the bits
// of every other kernel here.
//
// The loader needs no per-piece address registers: a piece is eight consecutive A rows (or 256 / BN k-rows of B), so a
// lane's offset inside a piece is one VGPR per image and the piece's position is a scalar offset.
//
// Chained segments.  A persistent stream-K workgroup runs several (tile, K-range) segments back to back; K2L starts each
// with an empty pipeline.  The loader instead walks ONE stream of slices through the ring: while the consumers finish
// a segment's last two slices it already fetches the next segment's first two, the fragment reads at the end of the
// last slice are the next segment's first, and the consumers' C / partial-tile stores -- and the drain that precedes a
// publish, which now waits for THEIR stores only -- run under loads in flight.  The ring position a segment starts at
// is then a run-time value: up to two slices run from copies of the slice body in front of the unrolled ring loop.
#pragma once
#include <type_traits>

#include "sgemm_dma.hpp"
#include "sgemm_mfma.hpp"   // the stream-K hand-over words (SK_*), GROUP_M

namespace mmh {

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
struct Dma5Tile {
  using T4 = DmaTile<BM, BN, KB, WTM, WTN, NBUF>;   // the consumers' geometry is K2L's
  static_assert(T4::WAVES == 4, "four consumer waves, one per SIMD");
  static constexpr int CONSUMERS = 4, LOADER = 4;   // wave index of the loader
  static constexpr int THREADS = 64 * 5;
  static constexpr int A_FLOATS = T4::A_FLOATS, B_FLOATS = T4::B_FLOATS, STAGE = T4::STAGE, KS = T4::KS;
  static constexpr int CHA = T4::CHA, CHB = T4::CHB, NP = CHA + CHB;   // 1 KiB pieces per slice
  static constexpr int RPC_A = T4::RPC_A, LPR_A = T4::LPR_A, RPC_B = T4::RPC_B, LPR_B = T4::LPR_B;
  static constexpr int LA = NBUF - 1;
  static_assert(NP <= 63, "vmcnt is a 6-bit counter");
  static_assert(RPC_B % 2 == 0, "the half-swap of odd k-rows must not depend on the piece");
  static constexpr size_t RING_BYTES = (size_t)NBUF * STAGE * sizeof(float);
  static constexpr size_t LDS_BYTES = RING_BYTES + 64;   // + the line the stream-K body passes a word through
};

// what follows a segment in its workgroup's stream, and the state carried from segment to segment (see sgemm_dma32.hpp)
struct Dma5Next {
  int tm = 0, tn = 0, kb = 0, len = 0;
};
struct Dma5Link {
  int pos = 0;
  bool primed = false;
};

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool PART_WT = false, bool EDGE = false, bool CHAIN = false>
struct Dma5Segment {
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF>;
  typedef float bfrag_t __attribute__((ext_vector_type(WTN)));
  typedef float afrag_t __attribute__((ext_vector_type(WTM)));
  typedef float c_vec_u __attribute__((ext_vector_type(WTN), aligned(4)));
  using c_vec = std::conditional_t<EDGE, c_vec_u, bfrag_t>;
  static constexpr int D = 2;   // fragments are read D k-steps ahead of the MFMAs that use them (ring of four sets)

  struct Frags {
    afrag_t a[4];
    bfrag_t b[4];
  };

  // per-lane constants: the consumers' fragment addresses, the loader's offsets inside a piece
  struct Lane {
    int wave, wm, wn, li, kq;
    bool loader;
    int a_off[8], b_off;
    uint32_t voff_a, voff_b;
    __device__ __forceinline__ void init(int lda, int ldb) {
      const int tid = threadIdx.x, lane = tid & 63;
      wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      loader = wave == T::LOADER;
      wm = wave / T::T4::WAVES_N;
      wn = wave % T::T4::WAVES_N;
      li = lane & 15;
      kq = lane >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) a_off[j] = (wm * 16 * WTM + li) * KB + 4 * (j ^ (li & 7)) + kq;
      b_off = WTN == 4 ? T::A_FLOATS + kq * BN + wn * 64 + 4 * li
                       : T::A_FLOATS + kq * BN + 4 * ((wn * 8 + (li >> 1)) ^ ((kq & 1) << 3)) + 2 * (li & 1);
      // loader: the 16-byte chunk a lane fetches is the one that belongs at its (swizzled) position of the image
      {
        const int r = lane / T::LPR_A, p = lane % T::LPR_A;                // piece j holds A rows RPC_A j + r
        voff_a = (uint32_t)(r * lda + 4 * (p ^ (r & 7))) * 4u;             // (RPC_A = 8: the row's swizzle is r's)
      }
      {
        const int r = lane / T::LPR_B, p = lane % T::LPR_B;                // piece j holds k-rows RPC_B j + r
        voff_b = (uint32_t)(r * ldb + 4 * (WTN == 2 ? (p ^ ((r & 1) << 3)) : p)) * 4u;
      }
    }
  };

  static __device__ __forceinline__ void run(float *lds, const Lane &L, int m, int n, int k, const float *__restrict__ A,
                                             int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                             int tm, int tn, int kb, int ke, bool init_from_c, const float *part_in,
                                             float *part_out, Frags &fr, Dma5Link &link, const Dma5Next nx = Dma5Next{}) {
    constexpr int KS = T::KS, STAGE = T::STAGE, A_FLOATS = T::A_FLOATS, NP = T::NP, LA = T::LA;
    static_assert(KB == 32 && T::RPC_A == 8, "a piece of A is eight rows of a 32-deep slice");
    static_assert(KS % 4 == 0 && KS > 2 * D, "fragment slots are numbered by k-step mod 4");
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
      if constexpr (CHAIN) __syncthreads();   // every wave is past its last fragment read of whatever ran before
      pos = 0;
      if constexpr (CHAIN) link.pos = n_slices % NBUF;
    }

    if (L.loader) {
      // ------------------------------------------------------------------ the loader wave
      // descriptors as base + extent SCALARS, packed where they are used (sgemm_dma32.hpp: a select between two
      // 128-bit descriptors goes through scratch memory)
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
      auto issue = [&](float *buf, int kt) {   // all NP pieces of stream slice kt into ring buffer `buf`
        const bool own = kt < ke;
        const int ks = own ? kt : kt + kdelta;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(own ? own_pa : next_pa), 0,
                                                                            own ? own_ea : next_ea, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(own ? own_pb : next_pb), 0,
                                                                            own ? own_eb : next_eb, 0x00020000);
        const uint32_t off_a = (uint32_t)(ks * KB) * 4u, off_b = (uint32_t)(ks * KB) * (uint32_t)ldb * 4u;
        static_for<T::CHA>([&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          DmaPiece::one(ra, buf + 256 * j, L.voff_a, off_a + (uint32_t)(T::RPC_A * j) * (uint32_t)lda * 4u);
        });
        static_for<T::CHB>([&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          DmaPiece::one(rb, buf + A_FLOATS + 256 * j, L.voff_b, off_b + (uint32_t)(T::RPC_B * j) * (uint32_t)ldb * 4u);
        });
      };
      if (!primed) {
        static_for<LA>([&](auto s_c) {
          constexpr int S = decltype(s_c)::value;
          issue(lds + S * STAGE, kb + S);
        });
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      int p2 = (pos + LA) % NBUF;   // ring position of stream slice kt + LA
      for (int kt = kb; kt < ke; ++kt) {
        issue(lds + p2 * STAGE, kt + LA);   // into the buffer slice kt - 1 was read from (its barrier is behind us)
        p2 = p2 == NBUF - 1 ? 0 : p2 + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NP) : "memory");   // stream slice kt + 1 is whole
        __builtin_amdgcn_s_barrier();
      }
      if (!chain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may still be landing in LDS
      return;
    }

    // -------------------------------------------------------------------- the consumer waves
    const int crow = row0 + L.wm * 16 * WTM + 4 * L.kq;
    const int ccol = col0 + L.wn * 16 * WTN + WTN * L.li;
    const bool whole_c = !EDGE || (rows_valid == BM && cols_valid == BN);
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
    auto frag_a = [&](const float *buf, auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      afrag_t a;
#pragma unroll
      for (int t = 0; t < WTM; ++t) a[t] = buf[L.a_off[ks & 7] + 4 * (ks & ~7) + t * 16 * KB];
      return a;
    };
    auto frag_b = [&](const float *buf, auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      return *reinterpret_cast<const bfrag_t *>(buf + L.b_off + 4 * ks * BN);
    };
    if (!primed) {
      __builtin_amdgcn_s_barrier();   // the loader has the first slice in LDS
      static_for<D>([&](auto d_c) {
        constexpr int d = decltype(d_c)::value;
        fr.a[d] = frag_a(lds, d_c);
        fr.b[d] = frag_b(lds, d_c);
      });
    }
    dma_stamp(1);

    // One K-slice out of ring buffer `buf` (K2L's slice body without its DMA pieces): per k-step the two fragment reads
    // for k-step ks + D, then the MFMAs of k-step ks; before k-step KS - D the slice's barrier -- from there on the reads go
    // to the NEXT buffer (the loader's counted wait says it is whole), and every read of this one has been issued.
    auto slice_at = [&](int kt, const float *buf, const float *nxt, auto tail_c) {
      constexpr bool TAIL = decltype(tail_c)::value;   // EDGE: the problem's last, partial slice
      const int krem = TAIL ? k - kt * KB : KB;
      static_for<KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks == KS - D) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (ks + D < KS) {
          fr.a[(ks + D) & 3] = frag_a(buf, std::integral_constant<int, ks + D>{});
          fr.b[(ks + D) & 3] = frag_b(buf, std::integral_constant<int, ks + D>{});
        } else {
          fr.a[(ks + D) & 3] = frag_a(nxt, std::integral_constant<int, ks + D - KS>{});
          fr.b[(ks + D) & 3] = frag_b(nxt, std::integral_constant<int, ks + D - KS>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        afrag_t a = fr.a[ks & 3];
        bfrag_t b = fr.b[ks & 3];
        if constexpr (TAIL) {
          // A's columns past k are the next row's floats or the caller's padding (NaN included): zero this lane's
          // operands of the k's that do not exist (B's rows there are zeros by descriptor; belt and braces)
          const bool live = 4 * ks + L.kq < krem;
#pragma unroll
          for (int t = 0; t < WTM; ++t) a[t] = live ? a[t] : 0.0f;
#pragma unroll
          for (int u = 0; u < WTN; ++u) b[u] = live ? b[u] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < WTM; ++t)
#pragma unroll
          for (int u = 0; u < WTN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    auto slice = [&](int kt, auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % NBUF;
      slice_at(kt, lds + CUR * STAGE, lds + NXT * STAGE, std::false_type{});
    };
    const int ke_main = ragged_k ? ke - 1 : ke;
    int kt = kb;
    // ONE exit per loop (with `break`s between the unrolled slices hipcc copies the accumulators on the hot path)
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
      if (ragged_k) {   // one slice per tile pays for a run-time ring position (an address add per fragment read)
        const int nx1 = pos == 2 ? 0 : pos + 1;
        slice_at(kt, lds + pos * STAGE, lds + nx1 * STAGE, std::true_type{});
      }
    }
    if (!chain) {
      // keep the fragments prefetched past the last slice formally alive (see sgemm_dma.hpp)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int u = 0; u < WTN; ++u) asm volatile("" ::"v"(fr.b[i][u]));
#pragma unroll
        for (int t = 0; t < WTM; ++t) asm volatile("" ::"v"(fr.a[i][t]));
      }
    }
    dma_stamp(2);

    auto out_vec = [&](int t, int r) {
      bfrag_t v;
#pragma unroll
      for (int u = 0; u < WTN; ++u) v[u] = acc[t][u][r];
      return v;
    };
    if (part_out) {
      __amdgpu_buffer_rsrc_t rsrc_p;
      if constexpr (PART_WT) rsrc_p = __builtin_amdgcn_make_buffer_rsrc(part_out, 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int t = 0; t < WTM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = crow + 16 * t + r;
          const bfrag_t v = out_vec(t, r);
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
        }
    } else if (whole_c) {
#pragma unroll
      for (int t = 0; t < WTM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<c_vec *>(C + (size_t)(crow + 16 * t + r) * ldc + ccol) = out_vec(t, r);
    } else {
#pragma unroll
      for (int t = 0; t < WTM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = crow + 16 * t + r;
          const bfrag_t v = out_vec(t, r);
          if (row < m) {
#pragma unroll
            for (int u = 0; u < WTN; ++u)
              if (ccol + u < n) C[(size_t)row * ldc + ccol + u] = v[u];
          }
        }
    }
  }
};

// One workgroup per C tile (XCD-aware block -> tile map), whole K range.
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE = false>
__global__ void __launch_bounds__(320)
sgemm_mfma_dma5_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                       float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using S = Dma5Segment<BM, BN, KB, WTM, WTN, NBUF, false, EDGE, false>;
  int tm, tn;
  dma_stamp(0);
  block_to_tile(blockIdx.x, nbm * nbn, nbm, nbn, tm, tn);
  typename S::Lane L;
  L.init(lda, ldb);
  typename S::Frags fr;
  Dma5Link link;
  S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, 0, (k + KB - 1) / KB, accumulate != 0, nullptr, nullptr, fr, link);
  dma_stamp_after_stores(3);
}

// ---------------------------------------------------------------------------------------------------------------
// K2Wp: the chained stream-K body.  Ranges, the order of a range's parts (head of the last tile FIRST, whole tiles, tail
// of the first tile LAST), the hand-over protocol and its words are streamk_body's (sgemm_mfma.hpp, K2p); the parts run
// as ONE stream of K-slices (Dma5Segment, CHAIN).  A tail's first slices are fetched BEFORE its hand-over word is
// looked at (they depend on nobody); should the word say the head's owner is not running (the wait-free path: leave),
// they are dropped.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE, bool CHAINED>
__device__ __forceinline__ void streamk5_body(float *lds, int m, int n, int k, const float *__restrict__ A, int lda,
                                              const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                              int accumulate, int nbm, int nbn, int *__restrict__ flags,
                                              float *__restrict__ parts, const int *__restrict__ order,
                                              const int *__restrict__ place, int *__restrict__ stats) {
  constexpr bool chained = CHAINED;
  using S = Dma5Segment<BM, BN, KB, WTM, WTN, NBUF, true, EDGE, true>;
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF>;
  const int nk = (k + KB - 1) / KB;
  const int Tn = nbm * nbn, G = gridDim.x;
  const int xcd = blockIdx.x % NXCD, local = blockIdx.x / NXCD;
  const int gq = G / NXCD, gr = G % NXCD;
  const int rho = (xcd < gr ? xcd * (gq + 1) : gr * (gq + 1) + (xcd - gr) * gq) + local;
  // (readfirstlane: what is loaded from memory or passed through LDS is workgroup-uniform, but hipcc cannot know -- and a
  // descriptor, LDS address or slice offset it takes for lane-dependent puts every LDS-DMA instruction into a waterfall loop)
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
  typename S::Frags fr;
  Dma5Link link;
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
    Dma5Next nx;
    if (chained && s + 1 < n_parts) {
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
        // ahead for this tail are dropped -- once they have landed (the loader's wait).
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
    S::run(lds, L, m, n, k, A, lda, B, ldb, C, ldc, tm, tn, pkb, pke, pkb == 0 && accumulate != 0, part_in,
           p.kind == HEAD ? my_slot : nullptr, fr, link, nx);
    if (p.kind == HEAD) {
      // Publish (cdna guide G16, recipe R1): the partial tile went out write-through (sc1) -- every storing wave
      // drains ITS stores (the loader's loads in flight are its own business), the workgroup meets, ONE lane ORs DONE in.
      if (!L.loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        head_reply = __hip_atomic_fetch_or(&flags[t_last], SK_HEAD_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// CHAINED = false: every part of a range starts with an empty pipeline (the A/B baseline, tools build)
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE = false, bool CHAINED = true>
__global__ void __launch_bounds__(320)
sgemm_dma5_streamk_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B,
                          int ldb, float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn,
                          int *__restrict__ flags, float *__restrict__ parts, const int *__restrict__ order,
                          const int *__restrict__ place, int *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  streamk5_body<BM, BN, KB, WTM, WTN, NBUF, EDGE, CHAINED>(lds, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nbm, nbn, flags,
                                                            parts, order, place, stats);
}

}  // namespace mmh
