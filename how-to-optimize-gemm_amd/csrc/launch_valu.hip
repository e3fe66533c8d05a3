// launch_valu.hip -- launchers of K1 (LDS-tiled, vector ALU only: BASELINE.json config 2) and K0 (naive).
// Part of libmmult_hip.so (see internal.hpp).
#include <cstdlib>

#include "launch_common.hpp"
#include "sgemm_valu.hpp"

namespace mmh {
namespace {

template <int BM, int BN, int KB, int NBUF, int P>
int launch_valu_tile_p(const GemmArgs &g) {
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  const bool fast = fast_shape(BM, BN, KB, g);
  constexpr size_t lds = (size_t)NBUF * (BM + BN) * KB * sizeof(float);
  dim3 grid((unsigned)(nbm * nbn)), block(256);
  if (fast)
    hipLaunchKernelGGL((sgemm_valu_kernel<BM, BN, KB, false, NBUF, P>), grid, block, lds, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb,
                       g.C, g.ldc, g.acc, nbm, nbn);
  else
    hipLaunchKernelGGL((sgemm_valu_kernel<BM, BN, KB, true, NBUF, P>), grid, block, lds, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb,
                       g.C, g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  char buf[160];
  snprintf(buf, sizeof buf, "sgemm_valu_kernel<%d,%d> K-slice %d x %d in LDS, fragments %d k-steps ahead, %s%d workgroups", BM, BN, KB,
           NBUF, P, fast ? "" : "guarded, ", nbm * nbn);
  set_last_launch(buf);
  return MMH_OK;
}

int valu_prefetch(int def) {   // A/B switch while measuring: MMH_VALU_P=1|2|4
  static const int v = [] { const char *e = std::getenv("MMH_VALU_P"); return e ? atoi(e) : 0; }();
  return (v == 1 || v == 2 || v == 4) ? v : def;
}

// fragment look-ahead: the 64x64 tile runs one wave per SIMD on its small shapes (32 cycles of FMAs per k-step against
// an LDS round trip of > 100), the 128x128 tile two (128 cycles per k-step each)
template <int BM, int BN, int KB, int NBUF>
int launch_valu_tile(const GemmArgs &g) {
  switch (valu_prefetch(BM == 64 ? 4 : 1)) {
    case 1: return launch_valu_tile_p<BM, BN, KB, NBUF, 1>(g);
    case 2: return launch_valu_tile_p<BM, BN, KB, NBUF, 2>(g);
    default: return launch_valu_tile_p<BM, BN, KB, NBUF, 4>(g);
  }
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
      // K1: the 128x128 rung, or the 64x64 tile where its rounds are cheaper -- a round of 64x64 tiles (one per CU;
      // ~76 TFLOP/s when the chip is full) costs 0.31 of a round of 128x128 tiles (~92): measured round 4, N = 1024 ..
      // 4096 step 256 (tools/tile_sweep.py --variants valu,valu_64x64,valu_128x128): 64x64 ahead at 1024 / 1280 / 1536 /
      // 2304 / 3072 (58 / 61 / 60 / 68 / 78 against 22 / 35 / 50 / 65 / 74), 128x128 elsewhere
      {
        const long tiles64 = (long)((g.m + 63) / 64) * ((g.n + 63) / 64);
        const long rounds64 = (tiles64 + cus - 1) / cus, rounds128 = (tiles128 + cus - 1) / cus;
        if (rounds64 * 10 < rounds128 * 32)
          return valu_nbuf() == 2 ? launch_valu_tile<64, 64, 64, 2>(g) : launch_valu_tile<64, 64, 64, 1>(g);
      }
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
