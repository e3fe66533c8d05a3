// launch_dma5.hip -- launchers of the LDS-DMA tiles with a loader wave (sgemm_dma5.hpp, K2W): 64x64, 128x64 and 128x128,
// each as one workgroup per tile or as the persistent stream-K form (chained parts), each in a whole-tile and a
// guarded (EDGE: any m, n, k, 4-byte aligned operands) instantiation.  Part of libmmult_hip.so (see internal.hpp).
#include "launch_common.hpp"
#include "sgemm_dma5.hpp"

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

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
int launch_dma5_tile(mmh_context *ctx, const GemmArgs &g) {
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF>;
  const int form = dma5_form<BM, BN, KB>(ctx, g);
  if (form < 0) return 1;
  const bool edge = form == 1;
  char what[224];
  if (ctx && ctx->streamk) {
    // the parts of a range as ONE stream of slices (MMH_OPT_STREAMK_CHAIN, default on), or each with a prologue of its own
    const bool chained = ctx->sk_chain != 0;
    auto kern = chained ? sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false, true>
                        : sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false, false>;
    auto kern_edge = chained ? sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true>
                             : sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, false>;
    auto occ = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true>;
    snprintf(what, sizeof what, "sgemm_dma5_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by a loader wave's LDS-DMA%s%s",
             BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, chained ? ", chained parts" : "", edge ? ", guarded" : "");
    const int sk = launch_streamk(ctx, edge ? kern_edge : kern, occ, BM, BN, KB, T::THREADS, T::LDS_BYTES, what, g);
    if (sk <= 0) return sk;
  }
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  auto kern = edge ? sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, true> : sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, false>;
  const int ok = allow_big_lds(kern, T::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B,
                     g.ldb, g.C, g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  snprintf(what, sizeof what,
           "sgemm_mfma_dma5_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by a loader wave's LDS-DMA, %s%d workgroups of %d threads",
           BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, edge ? "guarded, " : "", nbm * nbn, T::THREADS);
  set_last_launch(what);
  return MMH_OK;
}

template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
int warm_dma5_tile(mmh_context *ctx, float *scratch, hipStream_t s) {
  using T = Dma5Tile<BM, BN, KB, WTM, WTN, NBUF>;
  int rc;
  auto plain = [&](auto kern) {
    const int ok = allow_big_lds(kern, T::LDS_BYTES);
    if (ok != MMH_OK) return ok;
    hipLaunchKernelGGL(kern, dim3(1), dim3(T::THREADS), T::LDS_BYTES, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0,
                       1, 1);
    HIP_TRY(hipGetLastError());
    return (int)MMH_OK;
  };
  if ((rc = plain(sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, false>)) != MMH_OK) return rc;
  if ((rc = plain(sgemm_mfma_dma5_kernel<BM, BN, KB, WTM, WTN, NBUF, true>)) != MMH_OK) return rc;
  auto sk = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, false, true>;
  auto ske = sgemm_dma5_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF, true, true>;
  (void)resident_per_cu(ctx, ske, T::THREADS, T::LDS_BYTES);
  if ((rc = warm_streamk_kernel(sk, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s)) != MMH_OK) return rc;
  return warm_streamk_kernel(ske, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s);
}

}  // namespace

bool dma5_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA_64X64_DMA5: return dma5_form<64, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_128X64_DMA5: return dma5_form<128, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA_128X128_DMA5: return dma5_form<128, 128, 32>(ctx, g) >= 0;
    default: return false;
  }
}

int launch_dma5(mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA_64X64_DMA5:    // 64x64 tile, 4 consumer waves of 32x32 + the loader, 48 KiB ring: 3 workgroups per CU
      return launch_dma5_tile<64, 64, 32, 2, 2, 3>(ctx, g);
    case MMH_KERNEL_MFMA_128X64_DMA5:   // 128x64 tile, consumers of 64x32, 72 KiB ring: 2 per CU
      return launch_dma5_tile<128, 64, 32, 4, 2, 3>(ctx, g);
    case MMH_KERNEL_MFMA_128X128_DMA5:  // 128x128 tile, consumers of 64x64, 96 KiB ring
      return launch_dma5_tile<128, 128, 32, 4, 4, 3>(ctx, g);
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

int warm_dma5(mmh_context *ctx, float *scratch, hipStream_t s) {
  int rc;
  if ((rc = warm_dma5_tile<64, 64, 32, 2, 2, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  if ((rc = warm_dma5_tile<128, 64, 32, 4, 2, 3>(ctx, scratch, s)) != MMH_OK) return rc;
  return warm_dma5_tile<128, 128, 32, 4, 4, 3>(ctx, scratch, s);
}

}  // namespace mmh
