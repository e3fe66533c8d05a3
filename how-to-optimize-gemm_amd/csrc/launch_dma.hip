// launch_dma.hip -- launchers of the LDS-DMA tiles (sgemm_dma.hpp, K2L): 64x64, 128x64 and 128x128, each as one
// workgroup per tile or as the persistent stream-K form, each in a whole-tile (round 2) and a guarded (EDGE,
// round 3: any m, n, k, 4-byte aligned operands) instantiation.  Part of libmmult_hip.so (see internal.hpp).
#include "launch_common.hpp"
#include "sgemm_mfma.hpp"   // streamk_body (+ sgemm_dma.hpp)
#ifdef MMH_AB_BUILD
#include "sgemm_dma_rim.hpp"   // tools/ab/: round 3's rim (measured, it does not pay)
#endif

namespace mmh {
namespace {

// Which instantiation a shape runs: 0 = whole tiles, 16-byte aligned (the unguarded kernel), 1 = guarded,
// -1 = not on this family (descriptor window, or the guarded form is switched off / cannot take the rows).
template <int BM, int BN, int KB>
int dma_form(const mmh_context *ctx, const GemmArgs &g) {
  if (!window_ok(BM, BN, g.k, g.lda, g.ldb)) return -1;
  if (fast_shape(BM, BN, KB, g)) return 0;
  if (!ctx || !ctx->dma_edge) return -1;
  const bool rows16 = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && aligned16(g.A) && aligned16(g.B);
  if (!rows16 && !ctx->dma_dword_rows) return -1;
  return 1;
}

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
int launch_dma_tile(mmh_context *ctx, const GemmArgs &g) {
  using T = DmaTile<BM, BN, KB, WTM, WTN, NBUF>;
  const int form = dma_form<BM, BN, KB>(ctx, g);
  if (form < 0) return 1;
  const bool edge = form == 1;
  char what[224];
#ifndef MMH_AB_BUILD
  if (g.rim_m) return 1;   // (the rim is part of the tools build)
#else
  constexpr bool RIM_TILE = BN == 64 && ((BM == 64 && WTM == 2) || (BM == 128 && WTM == 4)) && WTN == 2;
  if (g.rim_m && !RIM_TILE) return 1;
  if constexpr (RIM_TILE) if (g.rim_m) {
    // tiles of the trimmed shape + the rim's workgroups in ONE plain launch.  Only where the trimmed shape would
    // be a plain launch anyway (the persistent stream-K grid owns every workgroup slot: nowhere for a rim to run)
    // and where every tile and every rim unit is resident from the start (see the kernel).
    const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
    const long tiles = (long)nbm * nbn;
    if (!ctx) return 1;
    auto occ = sgemm_dma_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true>;
    if (ctx->streamk && streamk_wanted(ctx, tiles, BM, BN, resident_per_cu(ctx, occ, T::THREADS, T::LDS_BYTES)) > 0) return 1;
    auto occ_rim = sgemm_mfma_dma_rim_kernel<BM, BN, KB, WTM, WTN, NBUF, true>;
    const int per_cu = resident_per_cu(ctx, occ_rim, T::THREADS, T::LDS_BYTES);
    const int rn = g.rim_n - g.n, rm = g.rim_m - g.m;
    // one rim workgroup per 16 elements: a column's 16 rows / a row's 16 columns
    const long rim_blocks = (long)rn * ((g.rim_m + 15) / 16) + (long)rm * (g.n / 16);
    if (tiles + rim_blocks > (long)per_cu * (ctx->cu_count > 0 ? ctx->cu_count : 256)) return 1;
    auto kern = edge ? sgemm_mfma_dma_rim_kernel<BM, BN, KB, WTM, WTN, NBUF, true> : sgemm_mfma_dma_rim_kernel<BM, BN, KB, WTM, WTN, NBUF, false>;
    const int ok = allow_big_lds(kern, T::LDS_BYTES);
    if (ok != MMH_OK) return ok;
    int acc_word = g.acc;
#ifdef MMH_AB_BUILD
    if (const char *ab = getenv("MMH_AB_RIM")) acc_word |= !strcmp(ab, "only") ? 2 : !strcmp(ab, "none") ? 4 : 0;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles + rim_blocks)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda,
                       g.B, g.ldb, g.C, g.ldc, acc_word, nbm, nbn, g.rim_m, g.rim_n);
    HIP_TRY(hipGetLastError());
    snprintf(what, sizeof what,
             "sgemm_mfma_dma_rim_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by LDS-DMA, %s%ld workgroups of %d "
             "threads on %d x %d + %ld on the rim (%d rows, %d columns, vector ALU)",
             BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, edge ? "guarded, " : "", tiles, T::THREADS, g.m, g.n, rim_blocks, rm, rn);
    set_last_launch(what);
    return MMH_OK;
  }
#endif
  if (ctx && ctx->streamk) {
    auto kern = sgemm_dma_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false>;
    auto kern_edge = sgemm_dma_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true>;
    snprintf(what, sizeof what, "sgemm_dma_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by LDS-DMA%s", BM,
             BN, 16 * WTM, 16 * WTN, KB, NBUF, edge ? ", guarded" : "");
    const int sk = launch_streamk(ctx, edge ? kern_edge : kern, kern_edge, BM, BN, KB, T::THREADS, T::LDS_BYTES, what, g);
    if (sk <= 0) return sk;
  }
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  auto kern = edge ? sgemm_mfma_dma_kernel<BM, BN, KB, WTM, WTN, NBUF, true> : sgemm_mfma_dma_kernel<BM, BN, KB, WTM, WTN, NBUF, false>;
  const int ok = allow_big_lds(kern, T::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B,
                     g.ldb, g.C, g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  snprintf(what, sizeof what,
           "sgemm_mfma_dma_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by LDS-DMA, %s%d workgroups of %d threads",
           BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, edge ? "guarded, " : "", nbm * nbn, T::THREADS);
  set_last_launch(what);
  return MMH_OK;
}

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
int warm_dma_tile(mmh_context *ctx, float *scratch, hipStream_t s) {
  using T = DmaTile<BM, BN, KB, WTM, WTN, NBUF>;
  int rc;
  auto plain = [&](auto kern) {
    const int ok = allow_big_lds(kern, T::LDS_BYTES);
    if (ok != MMH_OK) return ok;
    hipLaunchKernelGGL(kern, dim3(1), dim3(T::THREADS), T::LDS_BYTES, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0,
                       1, 1);
    HIP_TRY(hipGetLastError());
    return (int)MMH_OK;
  };
  if ((rc = plain(sgemm_mfma_dma_kernel<BM, BN, KB, WTM, WTN, NBUF, false>)) != MMH_OK) return rc;
  if ((rc = plain(sgemm_mfma_dma_kernel<BM, BN, KB, WTM, WTN, NBUF, true>)) != MMH_OK) return rc;
#ifdef MMH_AB_BUILD
  if constexpr (BN == 64) {   // the rim forms (AUTO only trims onto the 64-wide tiles)
    auto rim = [&](auto kern) {
      const int ok = allow_big_lds(kern, T::LDS_BYTES);
      if (ok != MMH_OK) return ok;
      hipLaunchKernelGGL(kern, dim3(2), dim3(T::THREADS), T::LDS_BYTES, s, BM, BN, KB, scratch, KB, scratch, BN + 1, scratch + 65536,
                         BN + 1, 0, 1, 1, BM, BN + 1);
      HIP_TRY(hipGetLastError());
      return (int)MMH_OK;
    };
    if ((rc = rim(sgemm_mfma_dma_rim_kernel<BM, BN, KB, WTM, WTN, NBUF, false>)) != MMH_OK) return rc;
    if ((rc = rim(sgemm_mfma_dma_rim_kernel<BM, BN, KB, WTM, WTN, NBUF, true>)) != MMH_OK) return rc;
  }
#endif
  auto sk = sgemm_dma_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false>;
  auto ske = sgemm_dma_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true>;
  (void)resident_per_cu(ctx, ske, T::THREADS, T::LDS_BYTES);
  if ((rc = warm_streamk_kernel(sk, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s)) != MMH_OK) return rc;
  return warm_streamk_kernel(ske, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s);
}

}  // namespace

bool dma_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA_64X64_DMA: return dma_form<64, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_128X64_DMA: return dma_form<128, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_128X128_DMA: return dma_form<128, 128, 32>(ctx, g) >= 0;
    default: return false;
  }
}

int launch_dma(mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA_64X64_DMA:   // 64x64 tile, 4 waves of 32x32, 48 KiB ring: 3 workgroups per CU
      return launch_dma_tile<64, 64, 32, 2, 2, 3>(ctx, g);
    case MMH_KERNEL_MFMA_128X64_DMA:  // 128x64 tile, 4 waves of 64x32, 72 KiB ring: 2 workgroups per CU
      return launch_dma_tile<128, 64, 32, 4, 2, 3>(ctx, g);
    case MMH_KERNEL_MFMA_128X128_DMA: // 128x128 tile, 4 waves of 64x64, 96 KiB ring
      return launch_dma_tile<128, 128, 32, 4, 4, 3>(ctx, g);
#ifdef MMH_AB_BUILD
    // A/B (valid results): the LDS-DMA tiles as EIGHT waves -- two waves per SIMD from one workgroup
    case 45: return launch_dma_tile<64, 64, 32, 1, 2, 3>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 46: return launch_dma_tile<128, 64, 32, 2, 2, 3>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    case 47: return launch_dma_tile<128, 128, 32, 4, 2, 3>(ctx, g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
#endif
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

int warm_dma(mmh_context *ctx, float *scratch, hipStream_t s) {
  int rc;
  if ((rc = warm_dma_tile<64, 64, 32, 2, 2, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma_tile<128, 64, 32, 4, 2, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  return warm_dma_tile<128, 128, 32, 4, 4, 3>(ctx, scratch, s);
}

#ifdef MMH_DMA_TIMELINE
// timeline build only: where the plain LDS-DMA kernels write their timeline stamps (4 x uint64 per workgroup;
// NULL switches them off).  tools/dma_timeline.py.
extern "C" int mmh_ab_set_stamps(mmh_handle_t h, void *stamps) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dma_stamps), &stamps, sizeof(void *)));
  return MMH_OK;
}
#endif

}  // namespace mmh
