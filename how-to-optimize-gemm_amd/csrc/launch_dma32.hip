// launch_dma32.hip -- launchers of the LDS-DMA tiles on v_mfma_f32_32x32x2_f32 (sgemm_dma32.hpp, K2M): 64x64, 128x64,
// 64x128 and 128x128, each as one workgroup per tile or as the persistent stream-K form (chained segments), each in a
// whole-tile and a guarded (EDGE: any m, n, k, 4-byte aligned operands) instantiation.  Part of libmmult_hip.so.
#include "launch_common.hpp"
#include "sgemm_dma32.hpp"
#include "sgemm_mfma.hpp"   // streamk_body

namespace mmh {

// ---- kernels ------------------------------------------------------------------------------------------------
// the UNCHAINED persistent form (streamk_body of sgemm_mfma.hpp over Dma32Seg): every segment its own prologue
template <int BM, int BN, int KB, int WM, int WN, int NBUF, bool EDGE = false>
__global__ void __launch_bounds__(256)
sgemm_dma32_streamk_kernel(int m, int n, int k, const float *__restrict__ A, int lda, const float *__restrict__ B,
                           int ldb, float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn,
                           int *__restrict__ flags, float *__restrict__ parts, const int *__restrict__ order,
                           const int *__restrict__ place, int *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  streamk_body<Dma32Seg<BM, BN, KB, WM, WN, NBUF, EDGE>>(lds, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nbm, nbn,
                                                          flags, parts, order, place, stats);
}

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

template <int BM, int BN, int KB, int WM, int WN, int NBUF>
int launch_dma32_tile(mmh_context *ctx, const GemmArgs &g) {
  using T = Dma32Tile<BM, BN, KB, WM, WN, NBUF>;
  const int form = dma32_form<BM, BN, KB>(ctx, g);
  if (form < 0) return 1;
  const bool edge = form == 1;
  char what[224];
  if (ctx && ctx->streamk) {
    auto kern = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, false>;
    auto kern_edge = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, true>;
    snprintf(what, sizeof what, "sgemm_dma32_streamk_kernel<%d,%d> wave tile %dx%d on 32x32x2, K-slice %d x %d ring buffers by LDS-DMA%s",
             BM, BN, 32 * WM, 32 * WN, KB, NBUF, edge ? ", guarded" : "");
    const int sk = launch_streamk(ctx, edge ? kern_edge : kern, kern_edge, BM, BN, KB, T::THREADS, T::LDS_BYTES, what, g);
    if (sk <= 0) return sk;
  }
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  auto kern = edge ? sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, true> : sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, false>;
  const int ok = allow_big_lds(kern, T::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(T::THREADS), T::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B,
                     g.ldb, g.C, g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  snprintf(what, sizeof what,
           "sgemm_mfma32_dma_kernel<%d,%d> wave tile %dx%d on 32x32x2, K-slice %d x %d ring buffers by LDS-DMA, %s%d workgroups of %d threads",
           BM, BN, 32 * WM, 32 * WN, KB, NBUF, edge ? "guarded, " : "", nbm * nbn, T::THREADS);
  set_last_launch(what);
  return MMH_OK;
}

template <int BM, int BN, int KB, int WM, int WN, int NBUF>
int warm_dma32_tile(mmh_context *ctx, float *scratch, hipStream_t s) {
  using T = Dma32Tile<BM, BN, KB, WM, WN, NBUF>;
  int rc;
  auto plain = [&](auto kern) {
    const int ok = allow_big_lds(kern, T::LDS_BYTES);
    if (ok != MMH_OK) return ok;
    hipLaunchKernelGGL(kern, dim3(1), dim3(T::THREADS), T::LDS_BYTES, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0,
                       1, 1);
    HIP_TRY(hipGetLastError());
    return (int)MMH_OK;
  };
  if ((rc = plain(sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, false>)) != MMH_OK) return rc;
  if ((rc = plain(sgemm_mfma32_dma_kernel<BM, BN, KB, WM, WN, NBUF, true>)) != MMH_OK) return rc;
  auto sk = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, false>;
  auto ske = sgemm_dma32_streamk_kernel<BM, BN, KB, WM, WN, NBUF, true>;
  (void)resident_per_cu(ctx, ske, T::THREADS, T::LDS_BYTES);
  if ((rc = warm_streamk_kernel(sk, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s)) != MMH_OK) return rc;
  return warm_streamk_kernel(ske, BM, BN, KB, T::THREADS, 160 * 1024, scratch, s);
}

}  // namespace

bool dma32_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA32_64X64_DMA: return dma32_form<64, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32_128X64_DMA: return dma32_form<128, 64, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32_64X128_DMA: return dma32_form<64, 128, 32>(ctx, g) >= 0;
    case MMH_KERNEL_MFMA32_128X128_DMA: return dma32_form<128, 128, 32>(ctx, g) >= 0;
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
  return warm_dma32_tile<128, 128, 32, 2, 2, 3>(ctx, scratch, s);
}

}  // namespace mmh
