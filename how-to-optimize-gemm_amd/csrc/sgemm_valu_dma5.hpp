// sgemm_valu_dma5.hpp -- K1W (round 5): the vector-ALU rung (BASELINE.json config 2, "LDS-tiled no-MFMA baseline";
// cuda/MMult_cuda_3.cu:10-53 ... cuda/MMult_cuda_9.cu:30-125) with K2W's loader waves.
//
// Why.  K1 (sgemm_valu.hpp) stages every K-slice through registers in its FMA stream: global loads, the k-major
// transpose of A (v_movs), eight ds_write_b128 and two barriers per slice sit between the v_pk_fma_f32 -- 58 / 83 / 92
// TFLOP/s at N = 1024 / 2048 / 4096 against a packed-FMA roof of 157 (profiles/r04_notes.md section 11).  Here the
// staging leaves the FMA waves as it left the MFMA waves in K2W (sgemm_dma5.hpp): NL loader waves write the ring of
// K-slice images by LDS-DMA (`buffer_load_dwordx4 ... lds`, counted vmcnt, one barrier per slice -- the same images,
// the same ring protocol), and the four consumer waves issue ds_read_b128 and v_pk_fma_f32 and nothing else.
//
// The images are K2W's (nothing can be transposed on its way through the DMA):
//   As[m][k]  row-major, 32 floats per row, the 16-byte chunk of row r at position p holding source chunk
//             p ^ ((r >> 1) & 1) (K2W XORs r & 7 for its one-float fragments; here ONE bit does -- below -- and costs the
//             consumers two address registers instead of sixteen);
//   Bs[k][n]  as it lies in memory (BN = 64: the halves of odd k-rows swapped, Dma5Tile::src_chunk_b).
// So a consumer thread reads A ALONG k -- one ds_read_b128 per owned row gives that row's values for FOUR k-steps -- and
// B per k-step as before (four consecutive columns per read).  A wave is 16 (tx) x 4 (t) threads: the sixteen tx of
// a t read the same A address (a broadcast) and sixteen consecutive chunks of B; the four t of a wave own rows
// 4 i + t of the wave's row block: consecutive rows alternate between the two halves of the 64 banks, and the rows of one
// half (t and t + 2) differ in bit 1, which the chunk XOR turns into different chunks -- four bank groups for the four
// addresses of a read.  Thread tile 8 x 8 (128x128 block, 64 accumulator pairs) or 4 x 4 (64x64 block).
//
// Per element C(i,j) one accumulator, one fused multiply-add per k in ascending k (v_pk_fma_f32 is two independent
// fmas): the chain of K1, K2 and the oracle -- identical bits, tests/test_gpu_round5.py.
//
// Whole tiles only (m, n multiples of the tile, k of 32, 16-byte aligned rows): everything else stays on K1's guarded
// instantiation.
//
// Round 6: the same consumer as a SEGMENT of K2W's machinery (Dma5ValuConsumer below, reached through
// Dma5Segment<..., VALU = true>): the loaders, the chained ring, the stream-K body with its hand-over words and partial
// tiles are sgemm_dma5.hpp's, only what happens between two barriers is this file's.  A ragged tile count then costs
// the vector-ALU rung what it costs the MFMA rung -- a head and a tail per workgroup -- instead of a whole idle round
// (N = 2176 on 128x128 tiles: 289 tiles for 256 CUs ran 56 TFLOP/s; the reference's own non-tensor rungs are flat across
// its sweep, cuda/output_MMult_cuda_3.m:5-29).  The partial sums travel as fp32 and each element's chain resumes where
// the head left it: the bits of the plain launch.
#pragma once
#include "sgemm_dma5.hpp"

namespace mmh {

template <int BM, int BN, int NBUF, int NL, int P, int AK>   // P: k-steps of B-fragment look-ahead; AK: k-steps per A read (4: ds_read_b128, 2: ds_read_b64)
struct ValuDma5 {
  using T = Dma5Tile<BM, BN, 32, BM / 32, BN / 32, NBUF, NL>;
  static_assert((BM == 64 || BM == 128) && (BN == 64 || BN == 128), "16 x 16 threads of 4x4 output blocks");
  static_assert(P >= 2 && P <= 3, "B fragments live in four register slots; the slice barrier sits at k-step 32 - P <= 30");
  static_assert(AK == 2 || AK == 4, "an A read is a row's values for two or four k-steps");
  static constexpr int KB = 32;
  static constexpr int TI = BM / 16, RJ = BN / 64, TJ = 4 * RJ;   // rows per thread, 4-column blocks per thread
  static constexpr int WROWS = BM / 4;                            // rows of a consumer wave's block
  static constexpr int THREADS = T::THREADS;
  static constexpr size_t LDS_BYTES = T::RING_BYTES;
};

// (WPE, the second launch bound: waves per SIMD the register allocation must leave room for -- two co-resident workgroups of
// the 128x128 tile with two loaders are 12 waves per CU, three per SIMD: 168 registers)
template <int BM, int BN, int NBUF, int NL, int P, int AK, int WPE>
__global__ void __launch_bounds__(64 * (4 + NL), WPE)
sgemm_valu_dma5_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                       float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn) {
  using V = ValuDma5<BM, BN, NBUF, NL, P, AK>;
  using T = typename V::T;
  constexpr int KB = 32, STAGE = T::STAGE, A_FLOATS = T::A_FLOATS, LA = T::LA, NPL = T::NPL;
  constexpr int TI = V::TI, RJ = V::RJ, TJ = V::TJ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  int tm, tn;
  block_to_tile_g_div(blockIdx.x, nbm * nbn, nbm, nbn, T::GM, tm, tn);
  const int row0 = tm * BM, col0 = tn * BN;
  const int nk = k / KB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  if (wave >= 4) {
    // ------------------------------------------------------------------ the loader waves (K2W's, whole tiles only)
    const int ld = wave - 4;
    uint32_t voff_a, voff_b[T::PB];
    {
      const int r = lane / 8, p = lane % 8;                              // piece j holds A rows 8 j + r
      voff_a = (uint32_t)(r * lda + 4 * (p ^ ((r >> 1) & 1))) * 4u;
    }
#pragma unroll
    for (int jj = 0; jj < T::PB; ++jj) {
      const int c = 64 * jj + lane, r = c / T::CPR_B, pc = c % T::CPR_B;   // piece PB g + jj holds k-rows RB g + r
      voff_b[jj] = (uint32_t)(r * ldb + 4 * T::src_chunk_b(r, pc)) * 4u;
    }
    const float *pa = A + (size_t)row0 * lda, *pb = B + col0;
    auto issue = [&](float *buf, int kt) {   // this loader's pieces of slice kt (past the end: the same instructions against empty descriptors)
      const uint32_t ext = kt < nk ? 0x7fffffffu : 0u;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pa), 0, ext, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pb), 0, ext, 0x00020000);
      const uint32_t off_a = (uint32_t)(kt * KB) * 4u, off_b = (uint32_t)(kt * KB) * (uint32_t)ldb * 4u;
      static_for<T::CHA / NL>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        const int j = NL * i + ld;
        DmaPiece::one(ra, buf + 256 * j, voff_a, off_a + (uint32_t)(8 * j) * (uint32_t)lda * 4u);
      });
      static_for<T::CHB / T::PB / NL>([&](auto i_c) {
        constexpr int i2 = decltype(i_c)::value;
        const int g = NL * i2 + ld;
        static_for<T::PB>([&](auto jj_c) {
          constexpr int jj = decltype(jj_c)::value;
          DmaPiece::one(rb, buf + A_FLOATS + 256 * (T::PB * g + jj), voff_b[jj], off_b + (uint32_t)(T::RB * g) * (uint32_t)ldb * 4u);
        });
      });
    };
    static_for<LA>([&](auto s_c) {
      constexpr int S = decltype(s_c)::value;
      issue(lds + S * STAGE, S);
    });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NPL) : "memory");
    __builtin_amdgcn_s_barrier();
    int p2 = LA % NBUF;
    for (int kt = 0; kt < nk; ++kt) {
      issue(lds + p2 * STAGE, kt + LA);   // into the buffer slice kt - 1 was read from (its barrier is behind us)
      p2 = p2 == NBUF - 1 ? 0 : p2 + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NPL) : "memory");   // slice kt + 1 is whole
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may still be landing in LDS
    return;
  }

  // ---------------------------------------------------------------------- the consumer waves
  const int tx = lane & 15, t = lane >> 4;
  const int wrow = V::WROWS * wave + t;           // this thread's rows: wrow + 4 i
  // A: row (wrow + 4 i), chunk c of the slice -> float offset (wrow + 4 i) * 32 + 4 (c ^ b), b = bit 1 of the row (the same
  // for every i): 8 (c >> 1) + 4 ((c & 1) ^ b) -- one base per parity of c, everything else an immediate
  const int a_bit = (wrow >> 1) & 1;
  const int a_even = wrow * KB + 4 * a_bit, a_odd = wrow * KB + 4 * (1 ^ a_bit);
  // B: k-row kk, this thread's 4-column block h -> A_FLOATS + kk * BN + 64 h + 4 tx (BN = 64: chunk tx ^ 8 on odd k-rows)
  const int b_col = A_FLOATS + 4 * tx, b_col_odd = A_FLOATS + 4 * (BN == 64 ? (tx ^ 8) : tx);

  f32x2 acc[TI][TJ / 2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int h = 0; h < RJ; ++h) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) v = *reinterpret_cast<const f32x4 *>(C + (size_t)(row0 + wrow + 4 * i) * ldc + col0 + 4 * tx + 64 * h);
      acc[i][2 * h] = f32x2{v[0], v[1]};
      acc[i][2 * h + 1] = f32x2{v[2], v[3]};
    }
  // the accumulators' initial values arrive before the K loop starts (see sgemm_valu.hpp)
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ / 2; j += 2) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1]));

  typedef float afrag_t __attribute__((ext_vector_type(AK)));
  afrag_t fa[2][TI];            // A: the current and the next group of AK k-steps
  f32x4 fb[4][RJ];              // B: k-step mod 4
  auto read_a = [&](const float *buf, auto g_c) __attribute__((always_inline)) {   // k-steps AK g .. AK g + AK - 1 of the slice in `buf` into set g mod 2
    constexpr int g = decltype(g_c)::value, c = AK * g / 4, half = AK * g % 4;
#pragma unroll
    for (int i = 0; i < TI; ++i)
      fa[g & 1][i] = *reinterpret_cast<const afrag_t *>(buf + ((c & 1) ? a_odd : a_even) + 4 * i * KB + 8 * (c >> 1) + half);
  };
  auto read_b = [&](const float *buf, auto kk_c) __attribute__((always_inline)) {   // k-row kk of the slice in `buf` into slot kk mod 4
    constexpr int kk = decltype(kk_c)::value;
#pragma unroll
    for (int h = 0; h < RJ; ++h) fb[kk % 4][h] = *reinterpret_cast<const f32x4 *>(buf + ((kk & 1) ? b_col_odd : b_col) + kk * BN + 64 * h);
  };

  __builtin_amdgcn_s_barrier();   // the loaders have the first slice in LDS
  read_a(lds, std::integral_constant<int, 0>{});
  static_for<P>([&](auto p_c) { read_b(lds, p_c); });

  // One K-slice out of ring buffer `buf`: per k-step the B fragment of k-step kk + P, two k-steps before its first use the next A group,
  // then the 2 TI RJ packed FMAs of k-step kk.  Before k-step KB - P the slice's barrier: every read of this buffer has
  // completed, the loaders' counted wait says the NEXT one is whole, and from here on the reads go there.
  // (always_inline: a slice body that stays a function of its own keeps every captured array in scratch memory)
  auto slice = [&](const float *buf, const float *nxt) __attribute__((always_inline)) {
    static_for<KB>([&](auto kk_c) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value, g = kk / AK, q = kk % AK;
      if constexpr (kk == KB - P) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (kk + P < KB) read_b(buf, std::integral_constant<int, kk + P>{});
      else read_b(nxt, std::integral_constant<int, kk + P - KB>{});
      if constexpr (q == AK - 2) {   // the next group, two k-steps before its first use
        if constexpr (g + 1 < KB / AK) read_a(buf, std::integral_constant<int, g + 1>{});
        else read_a(nxt, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const float a = fa[g & 1][i][q];
#pragma unroll
        for (int j = 0; j < TJ / 2; ++j) {
          const f32x4 bv = fb[kk % 4][j >> 1];
          const f32x2 b2 = (j & 1) ? f32x2{bv[2], bv[3]} : f32x2{bv[0], bv[1]};
          acc[i][j] = __builtin_elementwise_fma(f32x2{a, a}, b2, acc[i][j]);
        }
      }
      // pin the k-step (sgemm_valu.hpp: left alone hipcc hoists every read of the unrolled slice to its top)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ / 2; j += 2) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1])::"memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  static_assert(KB - P <= 30, "the next slice's first A chunk is requested at k-step 30: not before the slice's barrier");
  int kt = 0;
  while (kt + NBUF <= nk) {   // the ring unrolled: compile-time LDS offsets
    static_for<NBUF>([&](auto c_c) {
      constexpr int CUR = decltype(c_c)::value, NXT = (CUR + 1) % NBUF;
      slice(lds + CUR * STAGE, lds + NXT * STAGE);
    });
    kt += NBUF;
  }
  static_for<NBUF - 1>([&](auto c_c) {
    constexpr int CUR = decltype(c_c)::value, NXT = (CUR + 1) % NBUF;
    if (kt < nk) {
      slice(lds + CUR * STAGE, lds + NXT * STAGE);
      ++kt;
    }
  });
  // (the fragments requested past the last slice are never used; keep them formally alive so that their waits stay where they are)
#pragma unroll
  for (int i = 0; i < TI; ++i) asm volatile("" ::"v"(fa[0][i]));
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int h = 0; h < RJ; ++h) asm volatile("" ::"v"(fb[s][h]));

#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int h = 0; h < RJ; ++h) {
      const f32x4 v = {acc[i][2 * h][0], acc[i][2 * h][1], acc[i][2 * h + 1][0], acc[i][2 * h + 1][1]};
      *reinterpret_cast<f32x4 *>(C + (size_t)(row0 + wrow + 4 * i) * ldc + col0 + 4 * tx + 64 * h) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The vector-ALU consumer of one (tile, K-range) segment: Dma5Segment::consume's contract -- accumulators from zero / C
// / the head's partial tile, a first fragment read when the pipeline is not primed, one barrier per slice (the deferred
// publish of a stream-K head rides on the first), the ring entered at a run-time position, the result to C or, write-
// through, to the workgroup's partial-tile slot -- with the kernel above's k-step in between.
// ---------------------------------------------------------------------------------------------------------------
template <class S>
struct Dma5ValuConsumer {
  using T = typename S::T;
  static constexpr int BM = S::kBM, BN = S::kBN, NBUF = S::kNBUF, AK = S::kAK, KB = 32, P = 2;
  static constexpr int TI = BM / 16, RJ = BN / 64, TJ = 4 * RJ, STAGE = T::STAGE;
  static constexpr bool CHAIN = S::kChain;
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  static __device__ __forceinline__ void consume(float *lds, const typename S::Lane &L, float *__restrict__ C, int ldc, int row0,
                                                 int col0, int kb, int ke, int pos, bool primed, bool chain, bool init_from_c,
                                                 const float *part_in, float *part_out, typename S::Frags &fr, int *pub_flag,
                                                 int &pub_reply) {
    const int wrow = L.v_wrow, tx = L.v_tx;
    f32x2 acc[TI][TJ / 2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int h = 0; h < RJ; ++h) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (part_in) v = *reinterpret_cast<const f32x4 *>(part_in + (size_t)(wrow + 4 * i) * BN + 4 * tx + 64 * h);
        else if (init_from_c) v = *reinterpret_cast<const f32x4 *>(C + (size_t)(row0 + wrow + 4 * i) * ldc + col0 + 4 * tx + 64 * h);
        acc[i][2 * h] = f32x2{v[0], v[1]};
        acc[i][2 * h + 1] = f32x2{v[2], v[3]};
      }
    // the accumulators' initial values arrive before the K loop starts (see sgemm_valu.hpp)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ / 2; j += 2) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1]));

    typedef float afrag_t __attribute__((ext_vector_type(AK)));
    auto read_a = [&](const float *buf, auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value, c = AK * g / 4, half = AK * g % 4;
#pragma unroll
      for (int i = 0; i < TI; ++i)
        fr.a[g & 1][i] = *reinterpret_cast<const afrag_t *>(buf + ((c & 1) ? L.v_a_odd : L.v_a_even) + 4 * i * KB + 8 * (c >> 1) + half);
    };
    auto read_b = [&](const float *buf, auto kk_c) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value;
#pragma unroll
      for (int h = 0; h < RJ; ++h)
        fr.b[kk % 4][h] = *reinterpret_cast<const f32x4 *>(buf + ((kk & 1) ? L.v_b_col_odd : L.v_b_col) + kk * BN + 64 * h);
    };
    if (!primed) {
      __builtin_amdgcn_s_barrier();   // the loaders have the first slice in LDS
      read_a(lds, std::integral_constant<int, 0>{});
      static_for<P>([&](auto p_c) { read_b(lds, p_c); });
    }
    dma_stamp(L.stamp_base + 1);
    // a HEAD's deferred publish (sgemm_dma5.hpp): pending until the first slice barrier of this part
    bool pend = CHAIN && __builtin_amdgcn_readfirstlane((int)(pub_flag != nullptr)) != 0;
    auto slice_at = [&](const float *buf, const float *nxt) __attribute__((always_inline)) {
      static_for<KB>([&](auto kk_c) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_c)::value, g = kk / AK, q = kk % AK;
        if constexpr (kk == KB - P) {
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
        if constexpr (kk + P < KB) read_b(buf, std::integral_constant<int, kk + P>{});
        else read_b(nxt, std::integral_constant<int, kk + P - KB>{});
        if constexpr (q == AK - 2) {   // the next group, two k-steps before its first use
          if constexpr (g + 1 < KB / AK) read_a(buf, std::integral_constant<int, g + 1>{});
          else read_a(nxt, std::integral_constant<int, 0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const float a = fr.a[g & 1][i][q];
#pragma unroll
          for (int j = 0; j < TJ / 2; ++j) {
            const f32x4 bv = fr.b[kk % 4][j >> 1];
            const f32x2 b2 = (j & 1) ? f32x2{bv[2], bv[3]} : f32x2{bv[0], bv[1]};
            acc[i][j] = __builtin_elementwise_fma(f32x2{a, a}, b2, acc[i][j]);
          }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ / 2; j += 2) asm volatile("" : "+v"(acc[i][j]), "+v"(acc[i][j + 1])::"memory");
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    auto slice = [&](auto cur_c) __attribute__((always_inline)) {
      constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % NBUF;
      slice_at(lds + CUR * STAGE, lds + NXT * STAGE);
    };
    int kt = kb;
    if constexpr (CHAIN) {   // up to NBUF - 1 slices to reach ring position 0
      static_for<NBUF - 1>([&](auto p_c) {
        constexpr int PP = decltype(p_c)::value + 1;
        if (pos == PP && kt < ke) {
          slice(std::integral_constant<int, PP>{});
          ++kt;
          pos = (PP + 1) % NBUF;
        }
      });
    }
    while (kt + NBUF <= ke) {
      static_for<NBUF>([&](auto c_c) { slice(c_c); });
      kt += NBUF;
    }
    static_for<NBUF - 1>([&](auto c_c) {   // (only reached at ring position 0)
      if (kt < ke) {
        slice(c_c);
        ++kt;
      }
    });
    if (!chain) {
      // (the fragments requested past the last slice are never used; keep them formally alive so that their waits stay where they are)
#pragma unroll
      for (int i = 0; i < TI; ++i) asm volatile("" ::"v"(fr.a[0][i]));
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int h = 0; h < RJ; ++h) asm volatile("" ::"v"(fr.b[sl][h]));
    }
    dma_stamp(L.stamp_base + 2);
    __amdgpu_buffer_rsrc_t rsrc_p;
    if (part_out) rsrc_p = __builtin_amdgcn_make_buffer_rsrc(part_out, 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int h = 0; h < RJ; ++h) {
        const f32x4 v = {acc[i][2 * h][0], acc[i][2 * h][1], acc[i][2 * h + 1][0], acc[i][2 * h + 1][1]};
        if (part_out) {
          // the partial tile of a stream-K head: write-through (sc1), published by a drain + one atomic (sgemm_dma5.hpp)
          typedef int i32x4_t __attribute__((ext_vector_type(4)));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rsrc_p,
                                                 (uint32_t)(((wrow + 4 * i) * BN + 4 * tx + 64 * h) * 4), 0, 16);
        } else {
          *reinterpret_cast<f32x4 *>(C + (size_t)(row0 + wrow + 4 * i) * ldc + col0 + 4 * tx + 64 * h) = v;
        }
      }
  }
};

// K1Wp: the persistent stream-K launch of the vector-ALU rung -- sgemm_dma5_streamk_kernel with the consumer above.
template <int BM, int BN, int NBUF, int NL, int AK, int WPE>
__global__ void __launch_bounds__(64 * (4 + NL), WPE)
sgemm_valu_dma5_streamk_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                               float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn, int *__restrict__ flags,
                               float *__restrict__ parts, const int *__restrict__ order, const int *__restrict__ place,
                               int *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  asm volatile("" ::"s"(A), "s"(B), "s"(C), "s"(lda), "s"(ldb), "s"(ldc), "s"(k), "s"(flags), "s"(parts), "s"(order), "s"(place));
  streamk5_body<BM, BN, 32, BM / 32, BN / 32, NBUF, false, true, NL, AK, true>(lds, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nbm, nbn,
                                                                              flags, parts, order, place, stats);
}

}  // namespace mmh
