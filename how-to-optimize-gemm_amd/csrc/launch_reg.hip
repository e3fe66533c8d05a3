// launch_reg.hip -- launchers of the register-staged MFMA tiles (sgemm_mfma.hpp): one workgroup per tile,
// the persistent stream-K form, and the opt-in split-K form.  Part of libmmult_hip.so (see internal.hpp).
#include "launch_common.hpp"
#include "sgemm_mfma.hpp"

namespace mmh {
namespace {

// SIMPLE: the un-pipelined rung.  SCHED / BUFLD / ABL: see sgemm_mfma.hpp.  The
// buffer-descriptor path needs every byte offset inside a 2 GiB window; larger
// operands fall back to 64-bit global addressing (same kernel, BUFLD = false).
template <int BM, int BN, bool SIMPLE = false, int SCHED = 4, int ABL = 0, bool BUFLD = true, int WTN = 4,
          int WTM = 4, int KB = BK>
int launch_mfma(const GemmArgs &g) {
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  const bool fast = fast_shape(BM, BN, KB, g);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  dim3 grid((unsigned)(nbm * nbn)), block(threads);
  const bool win = window_ok(BM, BN, g.k, g.lda, g.ldb);
#define MMH_LAUNCH(KERN)                                                                   \
  do {                                                                                     \
    auto kern = KERN;                                                                      \
    const int ok = allow_big_lds(kern, lds);                                               \
    if (ok != MMH_OK) return ok;                                                           \
    hipLaunchKernelGGL(kern, grid, block, lds, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.acc, nbm, nbn); \
  } while (0)
  if constexpr (SIMPLE) {
    if (fast) MMH_LAUNCH((sgemm_mfma_simple_kernel<BM, BN, false>));
    else      MMH_LAUNCH((sgemm_mfma_simple_kernel<BM, BN, true>));
  } else if (!fast) {
    // guarded launch: buffer descriptors bound the reads (any alignment >= 4 B);
    // operands larger than the descriptor window use the per-element path
    if (BUFLD && win) MMH_LAUNCH((sgemm_mfma_kernel<BM, BN, true, SCHED, 0, true, WTN, WTM, KB>));
    else              MMH_LAUNCH((sgemm_mfma_kernel<BM, BN, true, SCHED, 0, false, WTN, WTM, KB>));
  } else if (BUFLD && win) {
    MMH_LAUNCH((sgemm_mfma_kernel<BM, BN, false, SCHED, ABL, BUFLD, WTN, WTM, KB>));
  } else {
    MMH_LAUNCH((sgemm_mfma_kernel<BM, BN, false, SCHED, ABL, false, WTN, WTM, KB>));
  }
#undef MMH_LAUNCH
  HIP_TRY(hipGetLastError());
  {
    char buf[160];
    snprintf(buf, sizeof buf, "%s<%d,%d> wave tile %dx%d, K-slice %d, %s%d workgroups of %d threads",
             SIMPLE ? "sgemm_mfma_simple_kernel" : "sgemm_mfma_kernel", BM, BN, 16 * WTM, 16 * WTN, KB,
             fast ? "" : "guarded, ", nbm * nbn, threads);
    set_last_launch(buf);
  }
  return MMH_OK;
}

// persistent stream-K launch of the register-staged tile config <BM, BN, WTN>
template <int BM, int BN, int WTN, int WTM = 4, int KB = BK>
int try_launch_streamk(mmh_context *ctx, const GemmArgs &g) {
  if (!ctx || !ctx->streamk) return 1;
  if (!window_ok(BM, BN, g.k, g.lda, g.ldb)) return 1;   // descriptor window
  // whole, 16-byte-aligned shapes run the unguarded kernel; everything else the guarded one (partial
  // tiles travel through a workspace, not through C, so C's alignment and ragged edges do not matter)
  const bool fast = fast_shape(BM, BN, KB, g);
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  auto kern_fast = sgemm_mfma_streamk_kernel<BM, BN, false, WTN, WTM, KB>;
  auto kern_edge = sgemm_mfma_streamk_kernel<BM, BN, true, WTN, WTM, KB>;
  char what[160];
  snprintf(what, sizeof what, "sgemm_mfma_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d%s", BM, BN, 16 * WTM,
           16 * WTN, KB, fast ? "" : ", guarded");
  return launch_streamk(ctx, fast ? kern_fast : kern_edge, kern_edge, BM, BN, KB, threads, lds, what, g);
}

// Opt-in split-K launch (sgemm_mfma.hpp, K2s) of tile config <BM, BN, WTN> with S concurrent K parts.
// Returns MMH_OK if it launched, 1 if the shape does not qualify.
template <int BM, int BN, int WTN, int WTM = 4, int KB = BK>
int try_launch_splitk(mmh_context *ctx, int S, const GemmArgs &g) {
  if (!ctx || !ctx->sticky_dev || S < 2) return 1;
  if (!window_ok(BM, BN, g.k, g.lda, g.ldb) || !fast_shape(BM, BN, KB, g)) return 1;
  const int nbm = g.m / BM, nbn = g.n / BN, nk = g.k / KB;
  const long tiles = (long)nbm * nbn;
  if (S > nk) S = nk;
  if (S < 2) return 1;
  const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  auto kern = sgemm_mfma_splitk_kernel<BM, BN, WTN, WTM, KB>;
  {
    const int ok = allow_big_lds(kern, lds);
    if (ok != MMH_OK) return ok;
  }
  const int per_cu = resident_per_cu(ctx, kern, threads, lds);
  while (S >= 2 && tiles * S > (long)per_cu * cus) --S;   // every part resident at once
  if (S < 2) return 1;
  int *flags = nullptr;
  float *parts = nullptr;
  const int rc = workspace_for(ctx, g.s, tiles, (size_t)tiles * (S - 1) * BM * BN * sizeof(float), &flags, &parts);
  if (rc != MMH_OK) return rc;
  if (ctx->fault) workspaces_suspect(ctx);   // a launch whose finishers time out leaves arrival counts behind
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * S)), dim3(threads), lds, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb, g.C,
                     g.ldc, g.acc, nbm, nbn, S, flags, ctx->sticky_dev, parts, ctx->spin_limit, ctx->fault);
  HIP_TRY(hipGetLastError());
  {
    char buf[176];
    snprintf(buf, sizeof buf,
             "sgemm_mfma_splitk_kernel<%d,%d> wave tile %dx%d, K-slice %d, %ld tiles x %d concurrent K parts",
             BM, BN, 16 * WTM, 16 * WTN, KB, tiles, S);
    set_last_launch(buf);
  }
  return MMH_OK;
}

// one tile of a plain register-staged instantiation, on scratch: code object loaded, LDS opted into
template <typename K>
int warm_plain(K kern, int BM, int BN, int KB, int threads, size_t lds, float *scratch, hipStream_t s) {
  const int ok = allow_big_lds(kern, lds);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0, 1, 1);
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

}  // namespace

int launch_reg_splitk(mmh_context *ctx, int bn, int S, const GemmArgs &g) {
  return bn == 64 ? try_launch_splitk<128, 64, 2>(ctx, S, g) : try_launch_splitk<128, 128, 4>(ctx, S, g);
}

int launch_reg(mmh_context *ctx, int kernel, const GemmArgs &g) {
  switch (kernel) {
    case MMH_KERNEL_MFMA_SIMPLE:
      return launch_mfma<128, 128, true>(g);
    case MMH_KERNEL_MFMA_PIPE:
      return launch_mfma<128, 128, false, 0, 0, false>(g);
    case MMH_KERNEL_MFMA_256:
      return launch_mfma<256, 128>(g);
    case MMH_KERNEL_MFMA: {
      // ragged tile counts go to the persistent stream-K launch (same arithmetic,
      // same bits); everything else is one workgroup per tile
      const int sk = try_launch_streamk<128, 128, 4>(ctx, g);
      if (sk <= 0) return sk;
      return launch_mfma<128, 128>(g);
    }
    case MMH_KERNEL_MFMA_TILES:   // K2 without stream-K (one workgroup per tile, always)
      return launch_mfma<128, 128>(g);
    case MMH_KERNEL_MFMA_256X256: {  // 256x256 tile, 8 waves of 128x64 (one workgroup per CU)
      const int sk = try_launch_streamk<256, 256, 4, 8, 32>(ctx, g);
      if (sk <= 0) return sk;
      return launch_mfma<256, 256, false, 4, 0, true, 4, 8, 32>(g);
    }
    case MMH_KERNEL_MFMA_64X64: {    // 64x64 tile, 4 waves of 32x32, 128-deep K-slices
      const int sk = try_launch_streamk<64, 64, 2, 2, 128>(ctx, g);
      if (sk <= 0) return sk;
      return launch_mfma<64, 64, false, 4, 0, true, 2, 2, 128>(g);
    }
    case MMH_KERNEL_MFMA_128X64: {   // 128x64 tile, 4 waves of 64x32
      const int sk = try_launch_streamk<128, 64, 2>(ctx, g);
      if (sk <= 0) return sk;
      return launch_mfma<128, 64, false, 4, 0, true, 2>(g);
    }
#ifdef MMH_AB_BUILD
    // ---- tools-only variants (libmmult_hip_ab.so); never part of the product library ----
    case 19: {  // A/B: B through LDS-DMA (buffer_load ... lds)
      const int nbm = g.m / 128, nbn = g.n / 128;
      if (!fast_shape(128, 128, 32, g)) return MMH_ERR_INVALID_ARG;
      auto kern = sgemm_mfma_kernel<128, 128, false, 4, 0, true, 4, 4, 32, true>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(256), lds_bytes(128, 128), g.s, g.m, g.n, g.k, g.A,
                         g.lda, g.B, g.ldb, g.C, g.ldc, g.acc, nbm, nbn);
      HIP_TRY(hipGetLastError());
      return MMH_OK;
    }
    // ablation builds of the 256x256 configuration (TIMING ONLY): 21 no global loads, 22 + no LDS
    // stores, 23 + no barrier, 24 + no fragment reads (MFMAs only)
    case 21: return launch_mfma<256, 256, false, 4, 1, true, 4, 8, 32>(g);
    case 22: return launch_mfma<256, 256, false, 4, 3, true, 4, 8, 32>(g);
    case 23: return launch_mfma<256, 256, false, 4, 7, true, 4, 8, 32>(g);
    case 24: return launch_mfma<256, 256, false, 4, 15, true, 4, 8, 32>(g);
    case 16: return launch_mfma<128, 128, false, 5>(g);   // staging cadence A/B: one op per 3 / 4 / 1 MFMAs instead of 2
    case 17: return launch_mfma<128, 128, false, 6>(g);
    case 18: return launch_mfma<128, 128, false, 7>(g);
    // Ablation builds of the shipping kernel (TIMING ONLY -- results are wrong):
    // 32 no global loads, 33 + no LDS stores, 34 + no barrier, 35 + no fragment reads.
    case 32: return launch_mfma<128, 128, false, 4, 1>(g);
    case 33: return launch_mfma<128, 128, false, 4, 3>(g);
    case 34: return launch_mfma<128, 128, false, 4, 7>(g);
    case 35: return launch_mfma<128, 128, false, 4, 15>(g);
    // the same for the 128x64 configuration (36 = loads always from the first two slices, i.e. cache-hot)
    case 36: return launch_mfma<128, 64, false, 4, 16, true, 2>(g);
    case 37: return launch_mfma<128, 64, false, 4, 1, true, 2>(g);
    case 38: return launch_mfma<128, 64, false, 4, 3, true, 2>(g);
    case 39: return launch_mfma<128, 64, false, 4, 7, true, 2>(g);
    case 40: return launch_mfma<128, 64, false, 4, 15, true, 2>(g);
    // and for the 64x64 configuration (one wave per SIMD, 128-deep slices)
    case 41: return launch_mfma<64, 64, false, 4, 1, true, 2, 2, 128>(g);
    case 42: return launch_mfma<64, 64, false, 4, 3, true, 2, 2, 128>(g);
    case 43: return launch_mfma<64, 64, false, 4, 7, true, 2, 2, 128>(g);
    case 44: return launch_mfma<64, 64, false, 4, 15, true, 2, 2, 128>(g);
#endif
    default:
      set_last_error("unknown kernel variant");
      return MMH_ERR_INVALID_ARG;
  }
}

// What MMH_KERNEL_AUTO can reach of this family, run once on one tile of scratch (and the residency of
// the persistent forms asked for): a first real launch then loads nothing and sets no attribute.
int warm_reg(mmh_context *ctx, float *scratch, hipStream_t s) {
  int rc;
#define WARM_TILE(BM, BN, WTN, WTM, KB)                                                                                  \
  do {                                                                                                                   \
    constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;                                                  \
    constexpr size_t lds = lds_bytes(BM, BN, KB);                                                                        \
    if ((rc = warm_plain(sgemm_mfma_kernel<BM, BN, false, 4, 0, true, WTN, WTM, KB>, BM, BN, KB, threads, lds, scratch, \
                         s)) != MMH_OK)                                                                                  \
      return rc;                                                                                                         \
    if ((rc = warm_plain(sgemm_mfma_kernel<BM, BN, true, 4, 0, true, WTN, WTM, KB>, BM, BN, KB, threads, lds, scratch,  \
                         s)) != MMH_OK)                                                                                  \
      return rc;                                                                                                         \
    auto sk = sgemm_mfma_streamk_kernel<BM, BN, false, WTN, WTM, KB>;                                                    \
    auto ske = sgemm_mfma_streamk_kernel<BM, BN, true, WTN, WTM, KB>;                                                    \
    (void)resident_per_cu(ctx, ske, threads, lds);                                                                       \
    if ((rc = warm_streamk_kernel(sk, BM, BN, KB, threads, 160 * 1024, scratch, s)) != MMH_OK) return rc;                \
    if ((rc = warm_streamk_kernel(ske, BM, BN, KB, threads, 160 * 1024, scratch, s)) != MMH_OK) return rc;               \
  } while (0)
  WARM_TILE(128, 128, 4, 4, 32);
  WARM_TILE(256, 256, 4, 8, 32);
  WARM_TILE(128, 64, 2, 4, 32);
  WARM_TILE(64, 64, 2, 2, 128);
#undef WARM_TILE
  return MMH_OK;
}

}  // namespace mmh
