// launch_dma5.hip -- launchers of the LDS-DMA tiles with loader waves (sgemm_dma5.hpp, K2W): 64x64, 128x64, 128x128 and
// the whole-round tiles 96x96 / 96x64 / 160x160 (160x96: tools build), each as one workgroup per tile or as the persistent stream-K form
// (chained parts), each in a whole-tile and a guarded (EDGE: any m, n, k, 4-byte aligned operands) instantiation.
// Part of libmmult_hip.so (see internal.hpp).
#include "launch_common.hpp"
#include "sgemm_dma5.hpp"
#ifdef MMH_AB_BUILD
#include "sgemm_dma5_rim.hpp"   // tools/ab/: round 4's fused rim (measured: 2.2x slower per edge tile)
#endif

namespace mmh {
namespace {

template <int BM, int BN, int KB>
int dma5_form(const mmh_context *ctx, const GemmArgs &g) {
  if (!window_ok(BM, BN, g.k, g.lda, g.ldb)) return -1;
  if (fast_shape(BM, BN, KB, g)) return 0;
  if (!ctx || !ctx->dma_edge) return -1;
  const bool rows16 = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && aligned16(g.A) && aligned16(g.B);
  if (!rows16 && !ctx->dma_dword_rows) return -1;
  return 1;
}

// SK: the tile has a stream-K form (the whole-round tiles are launched one workgroup per tile only)
template <int BM, int BN, int WTM, int WTN, int NBUF, int NL, int D, bool SK = true, int RS = 1>
int launch_dma5_tile(mmh_context *ctx, const GemmArgs &g) {
  constexpr int KB = 32;
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>;
  const int form = dma5_form<BM, BN, KB>(ctx, g);
  if (form < 0) return 1;
  const bool edge = form == 1;
  char what[320];
#ifdef MMH_AB_BUILD
  if constexpr (BM == 64 && BN == 64 && NL == 2 && SK) {
    // The RIM launch (tools build: measured slower than the thin edge tiles, sgemm_dma5.hpp rim_wave): one row and / or column past a 64-boundary rides on the trimmed shape's tiles (rim_wave,
    // sgemm_dma5.hpp) -- where the caller (MMH_KERNEL_AUTO's table, or the forced kernel's own rule on the TRIMMED
    // tile count) wants one workgroup per tile.
    int r_m = 0, r_n = 0;
    if (ctx && ctx->rim5 && edge && dma5_rim_dims(g.m, g.n, &r_m, &r_n)) {
      const int nbm0 = (g.m - r_m) / 64 + ((g.m - r_m) % 64 ? 1 : 0), nbn0 = (g.n - r_n) / 64 + ((g.n - r_n) % 64 ? 1 : 0);
      const long tiles0 = (long)nbm0 * nbn0;
      bool plain = g.form == 1;
      if (g.form == 0) {
        auto occ = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true, NL, D>;
        plain = !ctx->streamk || streamk_wanted(ctx, tiles0, BM, BN, resident_per_cu(ctx, occ, T::THREADS, T::LDS_BYTES)) == 0;
      }
      if (plain) {
        using SR = Dma5Segment<BM, BN, KB, WTM, WTN, NBUF, false, true, false, NL, D, true>;
        auto kern = sgemm_mfma_dma5_rim_kernel<BM, BN, KB, WTM, WTN, NBUF, NL, D>;
        const int ok = allow_big_lds(kern, SR::RIM_LDS_BYTES);
        if (ok != MMH_OK) return ok;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles0), dim3(T::THREADS + 64), SR::RIM_LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B,
                           g.ldb, g.C, g.ldc, g.acc, nbm0, nbn0, r_m, r_n);
        HIP_TRY(hipGetLastError());
        snprintf(what, sizeof what,
                 "sgemm_mfma_dma5_rim_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by %d loader waves' LDS-DMA, guarded, "
                 "%ld workgroups of %d threads on %d x %d + a rim wave for %d row(s), %d column(s) (vector ALU, out of the tiles' LDS)",
                 BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, NL, tiles0, T::THREADS + 64, g.m - r_m, g.n - r_n, r_m, r_n);
        set_last_launch(what);
        return MMH_OK;
      }
    }
  }
#endif
  GemmArgs ga = g;
#ifdef MMH_AB_BUILD   // A/B switches ride in the upper bits of `accumulate` (sgemm_dma5.hpp)
  if (ctx) ga.acc |= (ctx->ab_nodefer ? 2 : 0) | (ctx->ab_whole_ranges ? 4 : 0) | ((ctx->ab_group_m & 0xff) << 8);
#endif
  if constexpr (SK) {
    if (ctx && ctx->streamk) {
      // the parts of a range as ONE stream of slices (MMH_OPT_STREAMK_CHAIN, default on), or each with a prologue of its own
      const bool chained = ctx->sk_chain != 0;
      auto kern = chained ? sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false, true, NL, D, RS>
                          : sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false, false, NL, D, RS>;
      auto kern_edge = chained ? sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true, NL, D, RS>
                               : sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, false, NL, D, RS>;
      auto occ = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true, NL, D, RS>;
#ifdef MMH_AB_BUILD   // option 103: the residency of the instantiation that is launched (found on the CPU)
      if (ctx->ab_own_occ && !edge) occ = kern;
#endif
      snprintf(what, sizeof what,
               "sgemm_dma5_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by %d loader wave%s' LDS-DMA, fragments %d "
               "k-steps ahead%s%s",
               BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, NL, NL > 1 ? "s" : "", D, chained ? ", chained parts" : "", edge ? ", guarded" : "");
      // a thin last tile row / column (sgemm_mfma_dma5_kernel dispatches those last, at a fraction of a tile's cost) does
      // not make a tile count ragged: plain or persistent is decided on the whole tiles alone
      long decide = 0;
      if (edge) {
        const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
        const int thin_row = (nbm > 1 && g.m - (nbm - 1) * BM <= 16) ? 1 : 0, thin_col = (nbn > 1 && g.n - (nbn - 1) * BN <= 16) ? 1 : 0;
        if (thin_row || thin_col) decide = (long)(nbm - thin_row) * (nbn - thin_col);
      }
      const int sk = launch_streamk(ctx, edge ? kern_edge : kern, occ, BM, BN, KB, T::THREADS, T::LDS_BYTES, what, ga, decide,
                                     (BM == 128 && BN == 128) ? 10 : 0);   // (phase-ordered tables from one 128x128 tile per workgroup)
      if (sk <= 0) return sk;
    }
  }
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  auto kern = edge ? sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, true, NL, D, RS>
                   : sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, false, NL, D, RS>;
  const int ok = allow_big_lds(kern, T::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  // the tail split (sgemm_mfma_dma5_kernel): ONE whole round and a last round of JUST UNDER one tile per CU -- 0.85 CUs <
  // tiles - w CUs <= CUs: where the dispatcher was seen to pack (229 .. 256 of 256; a smaller last round spreads by itself,
  // and after two or more rounds the slots of a CU have drifted apart: splitting then only costs the overlap of the rounds,
  // -3 .. -15 % when forced) -- and K-slices enough that a second launch is small beside a tile (k >= 512): the last
  // round goes out as a launch of its own behind the whole one (dma5_tail_split, internal.hpp: the cost table prices it)
  const long tiles = (long)nbm * nbn;
  long first = tiles;
  if (ctx && ctx->split_tail) {
    const long cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    const long w = std::min(resident_per_cu(ctx, kern, T::THREADS, T::LDS_BYTES), 3);
    if (dma5_tail_split(tiles, w, cus, g.k) && (w * cus) % 8 == 0) first = w * cus;
  }
  const int acc_bits = edge ? g.acc : ga.acc;
  hipLaunchKernelGGL(kern, dim3((unsigned)first), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb, g.C,
                     g.ldc, acc_bits, nbm, nbn);
  if (first < tiles)
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles - first)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B,
                       g.ldb, g.C, g.ldc, acc_bits | (int)((unsigned)(first >> 3) << 16), nbm, nbn);
  HIP_TRY(hipGetLastError());
  snprintf(what, sizeof what,
           "sgemm_mfma_dma5_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by %d loader wave%s' LDS-DMA, fragments %d k-steps "
           "ahead, %s%d workgroups of %d threads%s",
           BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, NL, NL > 1 ? "s" : "", D, edge ? "guarded, " : "", nbm * nbn, T::THREADS,
           first < tiles ? " (the last round as a launch of its own)" : "");
  set_last_launch(what);
  return MMH_OK;
}

template <int BM, int BN, int WTM, int WTN, int NBUF, int NL, int D, bool SK = true>
int warm_dma5_tile(mmh_context *ctx, float *scratch, hipStream_t s) {
  constexpr int KB = 32;
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF, NL>;
  int rc;
  auto plain = [&](auto kern) {
    const int ok = allow_big_lds(kern, T::LDS_BYTES);
    if (ok != MMH_OK) return ok;
    hipLaunchKernelGGL(kern, dim3(1), dim3(T::THREADS), T::LDS_BYTES, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0,
                       1, 1);
    HIP_TRY(hipGetLastError());
    return (int)MMH_OK;
  };
  if ((rc = plain(sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, false, NL, D>)) != MMH_OK) return rc;
  if ((rc = plain(sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, true, NL, D>)) != MMH_OK) return rc;
  if constexpr (SK) {
    auto sk = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false, true, NL, D>;
    auto ske = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true, NL, D>;
    (void)resident_per_cu(ctx, ske, T::THREADS, T::LDS_BYTES);
    if ((rc = warm_streamk_kernel(sk, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s)) != MMH_OK) return rc;
    return warm_streamk_kernel(ske, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s);
  }
  return MMH_OK;
}

}  // namespace

bool dma5_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA_64X64_DMA5: return dma5_form<64, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_128X64_DMA5: return dma5_form<128, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_128X128_DMA5: return dma5_form<128, 128, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_96X96_DMA5: return dma5_form<96, 96, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_96X64_DMA5: return dma5_form<96, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_160X160_DMA5: return dma5_form<160, 160, 32>(ctx, g) >= 0;
    default: return false;
  }
}

int launch_dma5(mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    //                      BM   BN  WTM WTN NBUF NL D
    // (loader counts as measured, profiles/r04_notes.md: one loader wave on a consumer's SIMD holds that consumer back
    // -- and the workgroup, at every barrier; two or four spread the pieces -- 128x128 under chained stream-K at N = 2560:
    // 127.9 / 144.4 / 145.2 TFLOP/s with 1 / 2 / 4 loaders, 128x64 at N = 2432: 139.5 / 139.7 / 143.5)
    case MMH_KERNEL_MFMA_64X64_DMA5:    // 64x64 tile, 4 consumer waves of 32x32 + two loaders, 48 KiB ring: 3 workgroups per CU
      return launch_dma5_tile<64, 64, 2, 2, 3, 2, 2>(ctx, g);
    case MMH_KERNEL_MFMA_128X64_DMA5:   // 128x64 tile, consumers of 64x32 + four loaders, 72 KiB ring: 2 per CU
      return launch_dma5_tile<128, 64, 4, 2, 3, 4, 2>(ctx, g);
    case MMH_KERNEL_MFMA_128X128_DMA5:  // 128x128 tile, consumers of 64x64 + four loaders, 96 KiB ring
      return launch_dma5_tile<128, 128, 4, 4, 3, 4, 2>(ctx, g);
    case MMH_KERNEL_MFMA_96X96_DMA5:    // 96x96 tile, consumers of 48x48 (column-blocked B) + one loader, 72 KiB ring: 2 per CU
      return launch_dma5_tile<96, 96, 3, 3, 3, 1, 2, false>(ctx, g);
    case MMH_KERNEL_MFMA_96X64_DMA5:    // 96x64 tile, consumers of 48x32 + four loaders, 60 KiB ring: 2 per CU (round 5; N = 1152: 110.5 against 106.6 TFLOP/s)
      return launch_dma5_tile<96, 64, 3, 2, 3, 4, 2, false>(ctx, g);
    case MMH_KERNEL_MFMA_160X160_DMA5:  // 160x160 tile, consumers of 80x80 (column-blocked B) + four loaders, 120 KiB ring: 1 per CU (N = 2560: 256 of them)
      return launch_dma5_tile<160, 160, 5, 5, 3, 4, 2, false>(ctx, g);
#ifdef MMH_AB_BUILD
    // A/B (valid results): ONE loader wave (round 4's first form), and the 160-wide whole-round tiles that lost to the
    // chained stream-K launch of the 128-wide ones (N = 2560: 140.8 against 145.2; N = 1920 on 160x96: 130.9 against 137.5)
    case 64: return launch_dma5_tile<64, 64, 2, 2, 3, 1, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    // ring depth: one 64x64 workgroup per CU runs 0.43 us slices -- two slices of look-ahead are less than a DMA's latency
    case 65: return launch_dma5_tile<64, 64, 2, 2, 6, 2, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 66: return launch_dma5_tile<64, 64, 2, 2, 4, 2, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 67: return launch_dma5_tile<64, 64, 2, 2, 6, 4, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 68: return launch_dma5_tile<128, 64, 4, 2, 3, 1, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 69: return launch_dma5_tile<128, 64, 4, 2, 4, 4, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 72: return launch_dma5_tile<128, 128, 4, 4, 3, 1, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 79: return launch_dma5_tile<160, 96, 5, 3, 3, 1, 2, false>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 80: return launch_dma5_tile<160, 160, 5, 5, 3, 1, 2, false>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    // (rounds 4-5, fragment reads as a block: 1 / 2 / 4 loaders at N = 2560 -- 256 tiles, one whole round -- 139.9 / 139.8 /
    // 140.6 against the 128x128 tile's chained stream-K 144.6; four whole rounds at N = 5120: 143.9 against 150.2.  It was
    // the ten ds_read instructions per k-step leaving in one block (id 95 keeps that form); the four-loader form with
    // the reads spread is the product's MMH_KERNEL_MFMA_160X160_DMA5)
    case 82: return launch_dma5_tile<160, 160, 5, 5, 3, 2, 2, false>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    // round 5: 96x64 / 64x96 (N = 1152: 216 tiles -- one round of 256 CUs at 84 % -- instead of 324 tiles of 64x64 under stream-K)
    case 83: return launch_dma5_tile<96, 64, 3, 2, 3, 4, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;   // (with a stream-K form: never ahead)
    case 84: return launch_dma5_tile<96, 64, 3, 2, 3, 2, 2>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 85: return launch_dma5_tile<64, 96, 2, 3, 3, 2, 2, false>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    // round 6 (second session): the fragment reads as a BLOCK in front of the k-step's MFMAs (RS = 0: rounds 4-6's form) --
    // what every tile here ran until the reads were spread behind the first MFMAs (sgemm_dma5.hpp, RS)
    case 95: return launch_dma5_tile<160, 160, 5, 5, 3, 4, 2, false, 0>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 96: return launch_dma5_tile<128, 128, 4, 4, 3, 4, 2, true, 0>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 97: return launch_dma5_tile<128, 64, 4, 2, 3, 4, 2, true, 0>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 98: return launch_dma5_tile<64, 64, 2, 2, 3, 2, 2, true, 0>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 99: return launch_dma5_tile<96, 96, 3, 3, 3, 1, 2, false, 0>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    // (with the reads spread, the prefetch distance D = 1 / 2 / 3 k-steps measures the same on the 64x64, 96x64 and 128x64
    // tiles, N = 1024 .. 2048, 3072, 4096: +-0.3 % -- profiles/r06_prefetch_distance_ab.md)
    // (RS = 2, measured and dropped: the slice's barrier BEHIND the k-step's first MFMA, its wait in that instruction's 32
    // cycles of matrix-pipe time -- +-0.2 % on every tile and size: the barrier is not what the loop waits for)
#endif
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

int warm_dma5(mmh_context *ctx, float *scratch, hipStream_t s) {
  int rc;
#ifdef MMH_AB_BUILD
  {   // the RIM launch of the 64x64 tile: one tile + its rim (65 x 65 x 32 on scratch)
    using SR = Dma5Segment<64, 64, 32, 2, 2, 3, false, true, false, 2, 2, true>;
    auto kern = sgemm_mfma_dma5_rim_kernel<64, 64, 32, 2, 2, 3, 2, 2>;
    if ((rc = allow_big_lds(kern, SR::RIM_LDS_BYTES)) != MMH_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(1), dim3(448), SR::RIM_LDS_BYTES, s, 65, 65, 32, scratch, 32, scratch, 68, scratch + 65536, 68, 0, 1, 1,
                       1, 1);
    HIP_TRY(hipGetLastError());
  }
#endif
  if ((rc = warm_dma5_tile<64, 64, 2, 2, 3, 2, 2>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma5_tile<128, 64, 4, 2, 3, 4, 2>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma5_tile<128, 128, 4, 4, 3, 4, 2>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma5_tile<96, 96, 3, 3, 3, 1, 2, false>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma5_tile<160, 160, 5, 5, 3, 4, 2, false>(ctx, scratch, s)) != MMH_OK) return rc;
  return warm_dma5_tile<96, 64, 3, 2, 3, 4, 2, false>(ctx, scratch, s);
}

#ifdef MMH_DMA_TIMELINE
// timeline build only: where the plain K2W kernels write their timeline stamps (this translation unit's copy of
// g_dma_stamps; 4 x uint64 per workgroup; NULL switches them off).  tools/dma5_timeline.py.
extern "C" int mmh_ab_set_stamps5(mmh_handle_t h, void *stamps) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dma_stamps), &stamps, sizeof(void *)));
  return MMH_OK;
}
// 4 = the plain kernels' layout; 32 = the stream-K kernels' (sgemm_dma5.hpp, streamk5_body; tools/sk_timeline.py)
extern "C" int mmh_ab_set_stamp_stride5(mmh_handle_t h, int stride) {
  if (!h || (stride != 4 && stride != 32)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dma_stamp_stride), &stride, sizeof(int)));
  return MMH_OK;
}
#endif

}  // namespace mmh
