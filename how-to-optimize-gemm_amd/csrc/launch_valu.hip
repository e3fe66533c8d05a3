// launch_valu.hip -- launchers of K1 (LDS-tiled, vector ALU only: BASELINE.json config 2) and K0 (naive).
// Part of libmmult_hip.so (see internal.hpp).
#include <cstdlib>

#include "launch_common.hpp"
#include "sgemm_valu.hpp"

namespace mmh {
namespace {

template <int BM, int BN, int KB, int NBUF>
int launch_valu_tile(const GemmArgs &g) {
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  const bool fast = fast_shape(BM, BN, KB, g);
  constexpr size_t lds = (size_t)NBUF * (BM + BN) * KB * sizeof(float);
  dim3 grid((unsigned)(nbm * nbn)), block(256);
  if (fast)
    hipLaunchKernelGGL((sgemm_valu_kernel<BM, BN, KB, false, NBUF>), grid, block, lds, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb,
                       g.C, g.ldc, g.acc, nbm, nbn);
  else
    hipLaunchKernelGGL((sgemm_valu_kernel<BM, BN, KB, true, NBUF>), grid, block, lds, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb,
                       g.C, g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  char buf[96];
  snprintf(buf, sizeof buf, "sgemm_valu_kernel<%d,%d> K-slice %d x %d in LDS, %s%d workgroups", BM, BN, KB, NBUF,
           fast ? "" : "guarded, ", nbm * nbn);
  set_last_launch(buf);
  return MMH_OK;
}

int valu_nbuf() {   // A/B switch while measuring: MMH_VALU_NBUF=2 -> the double-buffered form
  static const int v = [] { const char *e = std::getenv("MMH_VALU_NBUF"); return e && *e == '2' ? 2 : 1; }();
  return v;
}

int launch_naive(const GemmArgs &g) {
  dim3 grid((unsigned)((g.n + 63) / 64), (unsigned)((g.m + 3) / 4)), block(256);
  hipLaunchKernelGGL(sgemm_naive_kernel, grid, block, 0, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.acc);
  HIP_TRY(hipGetLastError());
  set_last_launch("sgemm_naive_kernel");
  return MMH_OK;
}

}  // namespace

int launch_valu(mmh_context *ctx, int kernel, const GemmArgs &g) {
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const long tiles128 = (long)((g.m + 127) / 128) * ((g.n + 127) / 128);
  switch (kernel) {
    case MMH_KERNEL_VALU:
      // K1: the 128x128 rung from ~0.62 tiles per CU up, the 64x64 tile below (measured, N = 1024 .. 2048:
      // 33 / 54 TFLOP/s against 17 / 33 at N = 1024 / 1408, ahead at 1536 -- 144 tiles of 128x128 on 256 CUs --
      // and behind from 1664, 169 tiles)
      if (tiles128 * 100 < cus * 62) return valu_nbuf() == 2 ? launch_valu_tile<64, 64, 64, 2>(g) : launch_valu_tile<64, 64, 64, 1>(g);
      return valu_nbuf() == 2 ? launch_valu_tile<128, 128, 32, 2>(g) : launch_valu_tile<128, 128, 32, 1>(g);
    case MMH_KERNEL_VALU_128X128:
      return valu_nbuf() == 2 ? launch_valu_tile<128, 128, 32, 2>(g) : launch_valu_tile<128, 128, 32, 1>(g);
    case MMH_KERNEL_VALU_64X64:
      return valu_nbuf() == 2 ? launch_valu_tile<64, 64, 64, 2>(g) : launch_valu_tile<64, 64, 64, 1>(g);
    case MMH_KERNEL_NAIVE:
      return launch_naive(g);
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

int warm_valu(mmh_context *, float *, hipStream_t) { return MMH_OK; }   // not on MMH_KERNEL_AUTO's path

}  // namespace mmh
