// launch_valu.hip -- launchers of K1 (LDS-tiled, vector ALU only: BASELINE.json config 2) and K0 (naive).
// Part of libmmult_hip.so (see internal.hpp).
#include <cstdlib>

#include "launch_common.hpp"
#include "sgemm_valu.hpp"
#include "sgemm_valu_dma5.hpp"

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

// K1W (sgemm_valu_dma5.hpp, round 5): the same rung with the staging done by loader waves' LDS-DMA.  Whole tiles only:
// returns 1 when the shape is not one (the caller then launches K1's guarded instantiation).
template <int BM, int BN, int NBUF, int NL, int P, int AK, int WPE>
int launch_valu_w(const GemmArgs &g) {
  using V = ValuDma5<BM, BN, NBUF, NL, P, AK>;
  if (!fast_shape(BM, BN, 32, g) || !window_ok(BM, BN, g.k, g.lda, g.ldb)) return 1;
  const int nbm = g.m / BM, nbn = g.n / BN;
  auto kern = sgemm_valu_dma5_kernel<BM, BN, NBUF, NL, P, AK, WPE>;
  const int ok = allow_big_lds(kern, V::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(V::THREADS), V::LDS_BYTES, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb, g.C,
                     g.ldc, g.acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  char buf[224];
  snprintf(buf, sizeof buf,
           "sgemm_valu_dma5_kernel<%d,%d> thread tile %dx%d on v_pk_fma_f32, K-slice 32 x %d ring buffers by %d loader waves' LDS-DMA, "
           "%d workgroups of %d threads", BM, BN, BM / 16, BN / 16, NBUF, NL, nbm * nbn, V::THREADS);
  set_last_launch(buf);
  return MMH_OK;
}

// K1Wp (round 6): the persistent stream-K launch of a K1W tile for ragged tile counts -- K2W's body, ranges, hand-over
// words and workspaces (launch_streamk) around the vector-ALU consumer.  Returns 1 when the shape does not qualify or
// the launcher's rule prefers the plain launch.
template <int BM, int BN, int NBUF, int NL, int AK, int WPE>
int launch_valu_sk(mmh_context *ctx, const GemmArgs &g) {
  using T = Dma5Tile<BM, BN, 32, BM / 32, BN / 32, NBUF, NL>;
  if (!ctx || !ctx->streamk || !fast_shape(BM, BN, 32, g) || !window_ok(BM, BN, g.k, g.lda, g.ldb)) return 1;
  auto kern = sgemm_valu_dma5_streamk_kernel<BM, BN, NBUF, NL, AK, WPE>;
  char what[224];
  snprintf(what, sizeof what,
           "sgemm_valu_dma5_streamk_kernel<%d,%d> thread tile %dx%d on v_pk_fma_f32, K-slice 32 x %d ring buffers by %d loader waves' "
           "LDS-DMA, chained parts", BM, BN, BM / 16, BN / 16, NBUF, NL);
  return launch_streamk(ctx, kern, kern, BM, BN, 32, T::THREADS, T::LDS_BYTES, what, g);
}

int valu_w() {   // MMH_VALU_W=0: K1 as it was before round 5 (register-staged), for A/B
  static const int v = [] { const char *e = std::getenv("MMH_VALU_W"); return e && *e == '0' ? 0 : 1; }();
  return v;
}

// fragment look-ahead of the register-staged K1 (round 4, measured): the 64x64 tile runs one wave per SIMD on its small
// shapes (32 cycles of FMAs per k-step against an LDS round trip of > 100): four k-steps; the 128x128 tile two waves (128
// cycles per k-step each): one.  One LDS slice, the next one parked in registers (NBUF = 1).  (The P = 1 / 2 / 4 and
// NBUF = 2 instantiations behind MMH_VALU_P / MMH_VALU_NBUF left with round 5: K1W is the whole-tile path now.)
template <int BM, int BN, int KB>
int launch_valu_tile(const GemmArgs &g) {
  return launch_valu_tile_p<BM, BN, KB, 1, (BM == 64 ? 4 : 1)>(g);
}

// a K1 tile: K1W on whole-tile shapes, K1's own (guarded or not) instantiation otherwise
int launch_k1_128(mmh_context *ctx, const GemmArgs &g) {
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const long tiles = (long)((g.m + 127) / 128) * ((g.n + 127) / 128);
  if (valu_w() && !(ctx && ctx->ab_valu_old)) {
    const int sk = launch_valu_sk<128, 128, 2, 2, 2, 2>(ctx, g);   // (256 registers: ONE persistent workgroup per CU -- at 168 the chained body spills 36)
    if (sk <= 0) return sk;
  }
  // (from four tiles per CU the register-staged K1 is 1-3 % ahead -- N = 4096 / 6144: 92.9 / 94.4 against 91.8 / 91.9:
  // two loader waves per workgroup take issue slots from the FMA waves of two SIMDs and there is no round left to
  // fill; below that K1W leads by 8 % (2048) to 17 % (1536): profiles/r05_notes.md)
  if (valu_w() && !(ctx && ctx->ab_valu_old) && tiles < 4 * cus) {
    const int w = launch_valu_w<128, 128, 2, 2, 2, 2, 3>(g);   // 64 KiB ring, 168 registers: two workgroups per CU
    if (w <= 0) return w;
  }
  return launch_valu_tile<128, 128, 32>(g);
}
int launch_k1_128x64(mmh_context *ctx, const GemmArgs &g) {   // (K1W only: 72 KiB ring, 128 registers: two per CU)
  if (valu_w() && !(ctx && ctx->ab_valu_old)) {
    const int sk = launch_valu_sk<128, 64, 3, 2, 2, 3>(ctx, g);
    if (sk <= 0) return sk;
    const int w = launch_valu_w<128, 64, 3, 2, 2, 2, 3>(g);
    if (w <= 0) return w;
  }
  return launch_k1_128(ctx, g);
}
int launch_k1_64(mmh_context *ctx, const GemmArgs &g) {
  if (valu_w() && !(ctx && ctx->ab_valu_old)) {
    {
      const int sk = launch_valu_sk<64, 64, 3, 2, 2, 5>(ctx, g);
      if (sk <= 0) return sk;
    }
    // 48 KiB ring, 76-84 registers: three workgroups per CU.  A read as ds_read_b64 (two k-steps) where a CU holds ONE
    // workgroup -- one wave per SIMD: N = 1024 71.7 against 63.2 TFLOP/s -- as ds_read_b128 (four) otherwise (1536: 66.6
    // against 63.0, 4096: 81.0 against 77.5)
    const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
    const long tiles = (long)((g.m + 63) / 64) * ((g.n + 63) / 64);
    const int w = tiles <= cus ? launch_valu_w<64, 64, 3, 2, 2, 2, 5>(g) : launch_valu_w<64, 64, 3, 2, 2, 4, 5>(g);
    if (w <= 0) return w;
  }
  return launch_valu_tile<64, 64, 64>(g);
}

// MMH_KERNEL_VALU's tile and launch form.  Plain launches (round 5): a tile family runs at its chip-full rate R times the
// share of its last round that is filled, counted per CU -- (tiles / CUs) / ceil(tiles / CUs): a lone workgroup has its
// CU's vector ALUs to itself and runs about as fast as two sharing them, so what a ragged count costs is the CUs left
// idle while the fullest one finishes.  R measured at whole-round sizes: 93 (128x128), 86 (128x64), 80 (64x64) TFLOP/s.
// Persistent stream-K launches (round 6, profiles/r06_valu_sweep.md): no idle round, but ONE workgroup per CU where the
// tiles do not fill two -- one FMA wave per SIMD, whose roof is 77 TFLOP/s against 106 for two (mmh_probe_valu_f32):
//   128x128 (one per CU, 222 registers): 91 x t / (t + 0.17), t = tiles per workgroup (78.9 at 1.13 .. 89 at 3.5);
//   128x64 between one and two tiles per CU: 73;  64x64: 62 / 72 / 68 on one / two / three workgroups per CU.
// (128x64 on two workgroups per CU runs 60-75 and is left out.)  `form`: 1 = plain, 2 = persistent (GemmArgs::form).
int k1_pick_tile(const mmh_context *ctx, const GemmArgs &g, int *form) {
  const double cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const bool sk_ok = ctx && ctx->streamk;
  double best = -1.0;
  int best_kernel = MMH_KERNEL_VALU_64X64, best_form = 1;
  auto offer = [&](double est, int kernel, int f) {
    if (est > best) best = est, best_kernel = kernel, best_form = f;
  };
  auto family = [&](int bm, int bn, double rate, int kernel) {
    if (!fast_shape(bm, bn, 32, g)) return;   // K1W runs whole tiles only (the caller has checked 64x64)
    const double tiles = (double)(g.m / bm) * (g.n / bn), per_cu = tiles / cus;
    const double rounds = (double)(long)(per_cu + 0.999999);
    // (below one tile per CU every family's estimate is its rate x the share of the CUs it fills: the smaller tile wins
    // unless the larger one fills as many -- which it cannot)
    offer(rate * per_cu / (rounds < 1.0 ? 1.0 : rounds), kernel, 1);
    if (!sk_ok || tiles <= cus) return;
    const long w = (long)per_cu;   // whole workgroups per CU the tiles fill
    if (kernel == MMH_KERNEL_VALU_128X128) {
      if ((long)tiles % (long)cus != 0) offer(91.0 * per_cu / (per_cu + 0.17), kernel, 2);
    } else if (kernel == MMH_KERNEL_VALU_128X64) {
      if (w == 1) offer(73.0, kernel, 2);
    } else if ((long)tiles % ((w > 3 ? 3 : w) * (long)cus) != 0) {
      offer(w == 1 ? 62.0 : w == 2 ? 72.0 : 68.0, kernel, 2);
    }
  };
  family(128, 128, 93.0, MMH_KERNEL_VALU_128X128);
  family(128, 64, 86.0, MMH_KERNEL_VALU_128X64);
  family(64, 64, 80.0, MMH_KERNEL_VALU_64X64);
  *form = best_form;
  return best_kernel;
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
    case MMH_KERNEL_VALU:   // K1: tile and launch form by k1_pick_tile; whole-tile shapes on K1W, the others on K1's guarded kernels
      if (fast_shape(64, 64, 32, g)) {
        GemmArgs ga = g;
        const int pick = k1_pick_tile(ctx, g, &ga.form);
        switch (pick) {
          case MMH_KERNEL_VALU_128X128: return launch_k1_128(ctx, ga);
          case MMH_KERNEL_VALU_128X64: return launch_k1_128x64(ctx, ga);
          default: return launch_k1_64(ctx, ga);
        }
      }
      {   // ragged / unaligned: K1's guarded tiles, the 64x64 one where its rounds are cheaper (round 4's rule)
        const long tiles64 = (long)((g.m + 63) / 64) * ((g.n + 63) / 64);
        const long rounds64 = (tiles64 + cus - 1) / cus, rounds128 = (tiles128 + cus - 1) / cus;
        if (rounds64 * 10 < rounds128 * 32) return launch_k1_64(ctx, g);
      }
      return launch_k1_128(ctx, g);
    case MMH_KERNEL_VALU_128X128:
      return launch_k1_128(ctx, g);
    case MMH_KERNEL_VALU_64X64:
      return launch_k1_64(ctx, g);
    case MMH_KERNEL_VALU_128X64:
      return launch_k1_128x64(ctx, g);
    case MMH_KERNEL_NAIVE:
      return launch_naive(g);
#ifdef MMH_AB_BUILD   // K1W variants measured in round 5 (profiles/r05_notes.md): ring depth / loaders / A read width
    case 87: return launch_valu_w<128, 128, 3, 2, 2, 4, 2>(g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;   // one workgroup per CU, 96 KiB ring
    case 93: return launch_valu_w<128, 128, 2, 4, 2, 2, 3>(g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;   // four loaders
    case 94: return launch_valu_w<128, 128, 2, 1, 2, 2, 3>(g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;   // one loader
    case 89: return launch_valu_w<64, 64, 3, 2, 2, 2, 5>(g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;     // ds_read_b64 A reads whatever the tile count
    case 92: return launch_valu_w<64, 64, 3, 2, 2, 4, 5>(g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;     // ds_read_b128 A reads ...
    case 91: return launch_valu_w<64, 128, 3, 2, 2, 4, 3>(g) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;    // 64x128: level with 128x64
#endif
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

int warm_valu(mmh_context *, float *, hipStream_t) { return MMH_OK; }   // not on MMH_KERNEL_AUTO's path

}  // namespace mmh
