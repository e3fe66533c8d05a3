// launch_dma32.hip -- launchers of the LDS-DMA tiles on v_mfma_f32_32x32x2_f32 (sgemm_dma32.hpp, K2M): 64x64, 128x64,
// 64x128 and 128x128, each as one workgroup per tile or as the persistent stream-K form (chained segments), each in a
// whole-tile and a guarded (EDGE: any m, n, k, 4-byte aligned operands) instantiation.
// (Round 4: measured slower than the 16x16x4 tiles everywhere -- profiles/r04_notes.md -- so the family is part of the
// TOOLS build only: since round 5 it lives here, under tools/ab/, and only build_ab_library() compiles it, with
// -I csrc for the product's headers.)
#ifdef MMH_AB_BUILD
#include "launch_common.hpp"
#include "sgemm_dma32.hpp"
#include "sgemm_mfma.hpp"   // streamk_body

namespace mmh {

namespace {

template <int BM, int BN, int KB>
int dma32_form(const mmh_context *ctx, const GemmArgs &g) {
  if (!window_ok(BM, BN, g.k, g.lda, g.ldb)) return -1;
  if (fast_shape(BM, BN, KB, g)) return 0;
  if (!ctx || !ctx->dma_edge) return -1;
  const bool rows16 = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && aligned16(g.A) && aligned16(g.B);
  if (!rows16 && !ctx->dma_dword_rows) return -1;
  return 1;
}

template <int BM, int BN, int KB, int WM, int WN, int NBUF, int MB = 1>
int launch_dma32_tile(mmh_context *ctx, const GemmArgs &g) {
  using T = Dma32Tile<BM, BN, KB, WM, WN, NBUF, MB>;
  const int form = dma32_form<BM, BN, KB>(ctx, g);
  if (form < 0) return 1;
  const bool edge = form == 1;
  char what[224];
  if (ctx && ctx->streamk) {
    // the parts of a range as ONE stream of slices (chained: MMH_OPT_STREAMK_CHAIN, default on) or each with a
    // prologue of its own (the A/B baseline)
    const bool chained = ctx->sk_chain != 0;
    auto kern = chained ? sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, false, MB>
                        : sgemm_dma32_streamk_unchained_kernel<BM, BN, KB, WM, WN, NBUF, false, MB>;
    auto kern_edge = chained ? sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, true, MB>
                             : sgemm_dma32_streamk_unchained_kernel<BM, BN, KB, WM, WN, NBUF, true, MB>;
    auto occ = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, true, MB>;
    snprintf(what, sizeof what, "sgemm_dma32_streamk_kernel<%d,%d> wave tile %dx%d on %s, K-slice %d x %d ring buffers by LDS-DMA%s%s",
             BM, BN, 32 * MB * WM, 32 * WN, MB == 2 ? "32x32x1_2b" : "32x32x2", KB, NBUF, chained ? ", chained parts" : "", edge ? ", guarded" : "");
    const int sk = launch_streamk(ctx, edge ? kern_edge : kern, occ, BM, BN, KB, T::THREADS, T::LDS_BYTES, what, g);
    if (sk <= 0) return sk;
  }
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  auto kern = edge ? sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, true, 0, MB> : sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, false, 0, MB>;
  const int ok = allow_big_lds(kern, T::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B,
                     g.ldb, g.C, g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  snprintf(what, sizeof what,
           "sgemm_mfma32_dma_kernel<%d,%d> wave tile %dx%d on %s, K-slice %d x %d ring buffers by LDS-DMA, %s%d workgroups of %d threads",
           BM, BN, 32 * MB * WM, 32 * WN, MB == 2 ? "32x32x1_2b" : "32x32x2", KB, NBUF, edge ? "guarded, " : "", nbm * nbn, T::THREADS);
  set_last_launch(what);
  return MMH_OK;
}

template <int BM, int BN, int KB, int WM, int WN, int NBUF, int MB = 1>
int warm_dma32_tile(mmh_context *ctx, float *scratch, hipStream_t s) {
  using T = Dma32Tile<BM, BN, KB, WM, WN, NBUF, MB>;
  int rc;
  auto plain = [&](auto kern) {
    const int ok = allow_big_lds(kern, T::LDS_BYTES);
    if (ok != MMH_OK) return ok;
    hipLaunchKernelGGL(kern, dim3(1), dim3(T::THREADS), T::LDS_BYTES, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0,
                       1, 1);
    HIP_TRY(hipGetLastError());
    return (int)MMH_OK;
  };
  if ((rc = plain(sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, false, 0, MB>)) != MMH_OK) return rc;
  if ((rc = plain(sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, true, 0, MB>)) != MMH_OK) return rc;
  auto sk = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, false, MB>;
  auto ske = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, true, MB>;
  (void)resident_per_cu(ctx, ske, T::THREADS, T::LDS_BYTES);
  if ((rc = warm_streamk_kernel(sk, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s)) != MMH_OK) return rc;
  return warm_streamk_kernel(ske, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s);
  // (the unchained forms are an A/B switch: loaded on first use)
}

}  // namespace

bool dma32_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA32_64X64_DMA: return dma32_form<64, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32_128X64_DMA: return dma32_form<128, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32_64X128_DMA: return dma32_form<64, 128, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32_128X128_DMA: return dma32_form<128, 128, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32B_128X64_DMA: return dma32_form<128, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32B_64X128_DMA: return dma32_form<64, 128, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32B_128X128_DMA: return dma32_form<128, 128, 32>(ctx, g) >= 0;
    default: return false;
  }
}

int launch_dma32(mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA32_64X64_DMA:    // 64x64 tile, 4 waves of 32x32 (one 32x32x2 accumulator each), 48 KiB ring: 3 per CU
      return launch_dma32_tile<64, 64, 32, 1, 1, 3>(ctx, g);
    case MMH_KERNEL_MFMA32_128X64_DMA:   // 128x64 tile, 4 waves of 64x32, 72 KiB ring: 2 per CU
      return launch_dma32_tile<128, 64, 32, 2, 1, 3>(ctx, g);
    case MMH_KERNEL_MFMA32_64X128_DMA:   // 64x128 tile, 4 waves of 32x64 (8-byte B fragments and C stores), 72 KiB ring
      return launch_dma32_tile<64, 128, 32, 1, 2, 3>(ctx, g);
    case MMH_KERNEL_MFMA32_128X128_DMA:  // 128x128 tile, 4 waves of 64x64, 96 KiB ring
      return launch_dma32_tile<128, 128, 32, 2, 2, 3>(ctx, g);
    // the two-block form (v_mfma_f32_32x32x1_2b_f32): 64-row wave tiles, no swaps
    case MMH_KERNEL_MFMA32B_128X64_DMA:   // 4 waves of 64x32 as 2 x 2
      return launch_dma32_tile<128, 64, 32, 1, 1, 3, 2>(ctx, g);
    case MMH_KERNEL_MFMA32B_64X128_DMA:   // 4 waves of 64x32 as 1 x 4
      return launch_dma32_tile<64, 128, 32, 1, 1, 3, 2>(ctx, g);
    case MMH_KERNEL_MFMA32B_128X128_DMA:  // 4 waves of 64x64 as 2 x 2
      return launch_dma32_tile<128, 128, 32, 1, 2, 3, 2>(ctx, g);
#ifdef MMH_AB_BUILD
    // timing-only ablations of the plain launch (WRONG results): 52-55 the 128x64 tile, 56-59 the 64x64 tile,
    // each without swaps / without loop DMA / without A reads and swaps / MFMAs and barriers only
#define MMH_ABL(ID, BM_, BN_, WM_, WN_, A_)                                                                              \
    case ID: {                                                                                                           \
      using T = Dma32Tile<BM_, BN_, 32, WM_, WN_, 3>;                                                                    \
      auto kern = sgemm_mfma32_dma_kernel<BM_, BN_, 32, WM_, WN_, 3, false, A_>;                                         \
      const int ok = allow_big_lds(kern, T::LDS_BYTES);                                                                  \
      if (ok != MMH_OK) return ok;                                                                                       \
      const int nbm = g.m / BM_, nbn = g.n / BN_;                                                                        \
      hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, \
                         g.B, g.ldb, g.C, g.ldc, g.acc, nbm, nbn);                                                       \
      HIP_TRY(hipGetLastError());                                                                                        \
      set_last_launch("K2M ablation build");                                                                             \
      return MMH_OK;                                                                                                     \
    }
    MMH_ABL(52, 128, 64, 2, 1, 1)
    MMH_ABL(53, 128, 64, 2, 1, 2)
    MMH_ABL(54, 128, 64, 2, 1, 5)
    MMH_ABL(55, 128, 64, 2, 1, 15)
    MMH_ABL(56, 64, 64, 1, 1, 1)
    MMH_ABL(57, 64, 64, 1, 1, 2)
    MMH_ABL(58, 64, 64, 1, 1, 5)
    MMH_ABL(59, 64, 64, 1, 1, 15)
#undef MMH_ABL
#endif
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

int warm_dma32(mmh_context *ctx, float *scratch, hipStream_t s) {
  int rc;
  if ((rc = warm_dma32_tile<64, 64, 32, 1, 1, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma32_tile<128, 64, 32, 2, 1, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma32_tile<64, 128, 32, 1, 2, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma32_tile<128, 128, 32, 2, 2, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma32_tile<128, 64, 32, 1, 1, 3, 2>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma32_tile<64, 128, 32, 1, 1, 3, 2>(ctx, scratch, s)) != MMH_OK) return rc;
  return warm_dma32_tile<128, 128, 32, 1, 2, 3, 2>(ctx, scratch, s);
}

}  // namespace mmh
#endif
