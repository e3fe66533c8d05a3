// policy.hip -- sgemm_on(): argument checks, the empty contraction, and MMH_KERNEL_AUTO's tile choice (the
// reference's `NEW := MMult_xxx` makefile switch, cuda/makefile:1-3, as a run-time decision).  Pure host code;
// the kernels are launched by launch_reg.hip / launch_dma.hip / launch_valu.hip.
#include <algorithm>

#include "internal.hpp"
#include "launch_common.hpp"   // streamk_wanted

namespace mmh {

namespace {
constexpr int kSliceK = 32;   // K-slice depth of the MFMA tiles (sgemm_tile.hpp BK)

// MMH_KERNEL_AUTO: a COST TABLE, not thresholds (round 4; rounds 1-3 had hand-set rules fitted to one chip's sweep).
// Every candidate -- a tile family as a plain launch or as a persistent stream-K launch -- is priced in microseconds
// from what the shape makes of it, per CU:
//     plain:     t = fix_p + cmax * nk * s_p[min(cmax, w)]      cmax = ceil(tiles / CUs): the tiles of the fullest CU
//     stream-K:  t = fix_s[w'] + (tiles * nk / CUs) * s_s[w']   w'   = persistent workgroups per CU
// nk = ceil(k / 32) K-slices per tile, w the family's co-residency; s_x[o] is what a CU takes per tile-slice with o
// tiles co-resident, fix_x everything that does not scale with K (launch, pipeline fill, C store, hand-over).  The
// numbers are FITTED (tools/policy_fit.py, least squares in relative error) to a measured set of shapes x candidates
// (tools/policy_shapes.py, tools/tile_sweep.py) and live in policy_table.inc; a plain launch of more than one round
// of workgroups with a ragged last round is priced at the fit's own 90th-percentile residual (MMH_POLICY_MULTIROUND_MARGIN).  Everything is
// per CU, so the table serves any CU count.  profiles/r04_auto_regret.md: regret against the best measured candidate
// on 500 held-out shapes.  tools/calibrate_policy.sh regenerates the table on another box.
struct Family {
  int kernel, bm, bn, w, has_sk;
  int skw;   // persistent workgroups per CU of the family's stream-K launches (<= w: the guarded chained kernel's registers)
  float fix_p, s_p[3], fix_s[3], s_s[3];
  float fix_p_whole, fix_s_whole[3];   // the fixed costs of the whole-tile instantiation (no guards: fast_shape)
  float tile_p[3], tile_s[3];          // per tile (on the fullest CU / per CU's share), by occupancy: pipeline fill, C store
};
#include "policy_table.inc"
const Family kFamilies[] = {MMH_POLICY_FAMILIES};

struct Plan {
  int kernel = -1;
  int form = 0;   // 1 plain, 2 persistent stream-K (GemmArgs::form)
  double us = 0.0;
  int sk_w = 0;   // form 2: the persistent workgroups per CU the price was taken for (GemmArgs::sk_w)
  int bm = 0, bn = 0;
};

bool is_k2w(int kernel) {
  return kernel == MMH_KERNEL_MFMA_64X64_DMA5 || kernel == MMH_KERNEL_MFMA_128X64_DMA5 || kernel == MMH_KERNEL_MFMA_128X128_DMA5 ||
         kernel == MMH_KERNEL_MFMA_96X96_DMA5 || kernel == MMH_KERNEL_MFMA_96X64_DMA5 || kernel == MMH_KERNEL_MFMA_160X160_DMA5;
}
bool is_k2l(int kernel) {
  return kernel == MMH_KERNEL_MFMA_64X64_DMA || kernel == MMH_KERNEL_MFMA_128X64_DMA || kernel == MMH_KERNEL_MFMA_128X128_DMA;
}

Plan auto_plan_for(const mmh_context *ctx, const GemmArgs &g) {
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const double nk = (double)((g.k + kSliceK - 1) / kSliceK);
  Plan best;
  long tiles_rim = 0;
  bool any_dma5 = false;   // did any LDS-DMA family take the shape?
  for (const Family &f : kFamilies) {
    // (Rounds 2-3 kept K > 8192 -- B beyond the Infinity Cache, the config-4 panels -- on the 256x256 tile: K2L's small tiles
    // lost 1-6 % there.  K2W's do not: 2048 .. 4096 x 16384 x 16384 run 152.4-153.2 TFLOP/s on the 128x64 tile against
    // 150.2-150.3, and 2048 x 4096 x 16384 -- half a round of 256x256 tiles -- 151.5 against 74.8: the fence is gone, the
    // table decides; profiles/r04_big_k.md.)
    if (f.kernel != MMH_KERNEL_MFMA_256X256) {
      // (round 5: the K2L tiles are candidates too -- VERDICT r04 item 4: they shipped, AUTO never priced them, and on
      // small shapes their 256-thread workgroups and shorter preamble won by 5-8 %)
      if (is_k2w(f.kernel) ? !dma5_shape_ok(ctx, f.kernel, g) : !dma_shape_ok(ctx, f.kernel, g)) continue;
      any_dma5 = true;
    }
    long tiles = (long)((g.m + f.bm - 1) / f.bm) * ((g.n + f.bn - 1) / f.bn);
    {   // the 64x64 tile's RIM launch: one or two rows / columns past a 64-boundary cost no tiles of their own (plain only)
      int r_m = 0, r_n = 0;
      if (f.kernel == MMH_KERNEL_MFMA_64X64_DMA5 && ctx && ctx->rim5 && dma5_rim_dims(g.m, g.n, &r_m, &r_n))
        tiles_rim = (long)((g.m - r_m + 63) / 64) * ((g.n - r_n + 63) / 64);
      else
        tiles_rim = 0;
    }
    if (tiles_rim) tiles = tiles_rim;
    const long cmax = (tiles + cus - 1) / cus;
    const int occ = (int)std::min<long>(cmax, f.w);
    const bool whole = fast_shape(f.bm, f.bn, kSliceK, g);
    // K2W's thin edge tiles (a last tile row / column with at most 16 valid rows / columns: a fraction of a tile's MFMAs,
    // dispatched last, beside whole tiles) cost MMH_POLICY_THIN of a round where they add a tile to the fullest CU
    // (N = 1025 against 1024 on the 64x64 tile: 26.3 against 18.1 us; tools/policy_fit.py THIN)
    double cmax_p = (double)cmax;
    if (is_k2w(f.kernel) && !tiles_rim) {
      const int nbm = (g.m + f.bm - 1) / f.bm, nbn = (g.n + f.bn - 1) / f.bn;
      const int tr = (nbm > 1 && g.m - (nbm - 1) * f.bm <= 16) ? 1 : 0, tc = (nbn > 1 && g.n - (nbn - 1) * f.bn <= 16) ? 1 : 0;
      if (tr || tc) {
        const long full = (long)(nbm - tr) * (nbn - tc), cfull = (full + cus - 1) / cus;
        cmax_p = (double)cfull + MMH_POLICY_THIN * (double)(cmax - cfull);
      }
    }
    double t = (whole ? f.fix_p_whole : f.fix_p) + cmax_p * (nk * f.s_p[occ - 1] + f.tile_p[occ - 1]);
    if (cmax > f.w && tiles % ((long)f.w * cus) != 0) {   // a ragged last round
      // ... of between half a tile and one tile per CU: the dispatcher hands those out one per CU, or in pairs to the
      // CUs whose workgroups ended together -- a whole extra round (the same launch 142 and 110 TFLOP/s in two passes:
      // tools/policy_fit.py PAIRING); priced at the risk
      // (the launches launch_dma5.hip splits -- one round and a last round of just under a tile per CU -- are out of that class)
      const long rem = tiles % ((long)f.w * cus);
      const bool split = is_k2w(f.kernel) && (!ctx || ctx->split_tail) && dma5_tail_split(tiles, f.w, cus, g.k);
      t *= (2 * rem > cus && rem <= cus && !split) ? std::max(MMH_POLICY_PAIRING_MARGIN, MMH_POLICY_MULTIROUND_MARGIN) : MMH_POLICY_MULTIROUND_MARGIN;
    }
    if (best.kernel < 0 || t < best.us) best = Plan{f.kernel, 1, t, 0, f.bm, f.bn};
    if (f.has_sk && (!ctx || ctx->streamk) && !tiles_rim) {
      // the grid launch_streamk will launch: the largest w' <= skw workgroups per CU that leaves every one a whole tile
      // (round 4 took w' from the plain co-residency w and priced 768-tile launches that ran on 512)
      int wp = 0;
      for (int c = f.skw; c >= 1; --c)
        if (tiles >= (long)c * cus) { wp = c; break; }
      if (wp > 0 && tiles % ((long)wp * cus) != 0 && tiles <= (1L << 24)) {
        double ts = (whole ? f.fix_s_whole[wp - 1] : f.fix_s[wp - 1]) +
                    (double)tiles / (double)cus * (nk * f.s_s[wp - 1] + f.tile_s[wp - 1]);
        // the K2W 64x64 tile under stream-K: the one candidate whose rate moves 2-5 % from run to run (tools/policy_fit.py
        // T64SK): priced at the risk
        if (f.kernel == MMH_KERNEL_MFMA_64X64_DMA5 && wp >= 2) ts *= MMH_POLICY_T64SK_MARGIN;   // (two per CU: from 512 tiles)
        if (ts < best.us) best = Plan{f.kernel, 2, ts, wp, f.bm, f.bn};
      }
    }
  }
  // No LDS-DMA family takes the shape (operands beyond the 2 GiB descriptor window; MMH_OPT_DMA_EDGE = 0 on a ragged or
  // unaligned shape): the table's one remaining row -- the 256x256 tile, priced without rivals -- is not a choice.
  // fallback_kernel picks among the register-staged tiles by tile count (round 4 launched 16 workgroups of 256x256
  // for 1000^3 here).
  if (!any_dma5) return Plan{};
  return best;
}

// operands the descriptors cannot window (beyond 2 GiB), or the guarded LDS-DMA form switched off: the
// register-staged tiles, by how many tiles the shape has (rounds 1-2)
int fallback_kernel(const mmh_context *ctx, const GemmArgs &g) {
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const long tiles128 = (long)((g.m + 127) / 128) * ((g.n + 127) / 128);
  const long tiles128x64 = (long)((g.m + 127) / 128) * ((g.n + 63) / 64);
  const long tiles256 = (long)((g.m + 255) / 256) * ((g.n + 255) / 256);
  if (tiles256 >= cus) return MMH_KERNEL_MFMA_256X256;
  if (tiles128x64 * 2 <= cus) return MMH_KERNEL_MFMA_64X64;
  if (tiles128 * 10 < cus * 8) return MMH_KERNEL_MFMA_128X64;
  return MMH_KERNEL_MFMA;
}
}  // namespace

int sgemm_on(mmh_context *ctx, int kernel, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
             float *dC, int ldc, int accumulate, hipStream_t s) {
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) {
    set_last_error("invalid argument");
    return rc;
  }
  if (m == 0 || n == 0) return MMH_OK;
  if (k == 0) {
    // empty contraction: C = 0 (overwrite) or C unchanged (accumulate)
    if (!accumulate)
      HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float), (size_t)m, s));
    return MMH_OK;
  }
  const GemmArgs g{m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate ? 1 : 0, s};
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const long tiles128 = (long)((m + 127) / 128) * ((n + 127) / 128);
  if (kernel == MMH_KERNEL_AUTO) {
    // OPT-IN split-K (default off: it gives up the one-chain-per-element bits, see sgemm_mfma.hpp K2s):
    // shapes with fewer 128x128 tiles than workgroup slots run their K ranges concurrently
    const long tiles256 = (long)((m + 255) / 256) * ((n + 255) / 256);
    if (ctx && ctx->splitk > 0 && tiles128 < cus && tiles256 < cus) {
      int S = ctx->splitk;
      if (S == 1) {   // auto: fill two workgroups per CU, keep >= 8 K-slices per part
        S = (int)((2 * cus) / (tiles128 > 0 ? tiles128 : 1));
        S = std::min(std::min(S, k / (8 * kSliceK)), 8);
      }
      if (S >= 2) {
        const int sk = launch_reg_splitk(ctx, 128, S, g);
        if (sk <= 0) return sk;
      }
    }
#ifdef MMH_AB_BUILD
    // MMH_OPT_RIM (tools build; measured, it does not pay -- profiles/r03_notes.md section 6): a few rows / columns past
    // a 64-boundary: the K2L tiles take the trimmed shape, the rim runs on the vector ALU in the same launch
    if (ctx && ctx->rim > 0) {
      const int rm = m % 64, rn = n % 64;
      if ((rm || rn) && rm <= ctx->rim && rn <= ctx->rim && m - rm >= 256 && n - rn >= 256 && k >= 64) {
        GemmArgs t = g;
        t.m = m - rm;
        t.n = n - rn;
        t.rim_m = m;
        t.rim_n = n;
        const long t64 = (long)(t.m / 64) * (t.n / 64);
        const int k0 = t64 <= 3 * cus ? MMH_KERNEL_MFMA_64X64_DMA : MMH_KERNEL_MFMA_128X64_DMA;
        const int d = launch_dma(ctx, k0, t);
        if (d <= 0) return d;
      }
    }
#endif
    const Plan plan = auto_plan_for(ctx, g);
    if (plan.kernel >= 0) {
      GemmArgs ga = g;
      ga.form = plan.form;
      ga.sk_w = plan.sk_w;
      if (plan.kernel == MMH_KERNEL_MFMA_256X256) return launch_reg(ctx, plan.kernel, ga);
      const int d = is_k2l(plan.kernel) ? launch_dma(ctx, plan.kernel, ga) : launch_dma5(ctx, plan.kernel, ga);
      if (d <= 0) return d;
    }
    kernel = fallback_kernel(ctx, g);
  }
  switch (kernel) {
    case MMH_KERNEL_VALU:
    case MMH_KERNEL_VALU_128X128:
    case MMH_KERNEL_VALU_64X64:
    case MMH_KERNEL_VALU_128X64:
    case MMH_KERNEL_NAIVE:
      return launch_valu(ctx, kernel, g);
    case MMH_KERNEL_MFMA_64X64_DMA: {   // K2L; shapes it does not take run the register-staged tile of the same size
      const int d = launch_dma(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA_64X64, g);
    }
    case MMH_KERNEL_MFMA_128X64_DMA: {
      const int d = launch_dma(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA_128X64, g);
    }
    case MMH_KERNEL_MFMA_128X128_DMA: {
      const int d = launch_dma(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA, g);
    }
    case MMH_KERNEL_MFMA_64X64_DMA5: {   // K2W; shapes it does not take run the register-staged tile of the same size
      const int d = launch_dma5(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA_64X64, g);
    }
    case MMH_KERNEL_MFMA_128X64_DMA5: {
      const int d = launch_dma5(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA_128X64, g);
    }
    case MMH_KERNEL_MFMA_128X128_DMA5:
    case MMH_KERNEL_MFMA_96X64_DMA5:
    case MMH_KERNEL_MFMA_160X160_DMA5:
    case MMH_KERNEL_MFMA_96X96_DMA5: {
      const int d = launch_dma5(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA, g);
    }
#ifdef MMH_AB_BUILD
    case MMH_KERNEL_MFMA32_64X64_DMA: {   // K2M; shapes it does not take run the register-staged tile of the same size
      const int d = launch_dma32(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA_64X64, g);
    }
    case MMH_KERNEL_MFMA32_128X64_DMA:
    case MMH_KERNEL_MFMA32B_128X64_DMA:
    case MMH_KERNEL_MFMA32B_64X128_DMA:
    case MMH_KERNEL_MFMA32_64X128_DMA: {
      const int d = launch_dma32(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA_128X64, g);
    }
    case MMH_KERNEL_MFMA32B_128X128_DMA:
    case MMH_KERNEL_MFMA32_128X128_DMA: {
      const int d = launch_dma32(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA, g);
    }
#endif
    case MMH_KERNEL_MFMA_SPLITK: {   // K2s forced: 128x128 tiles, ctx->splitk parts (auto when <= 1)
      int S = ctx ? ctx->splitk : 0;
      if (S <= 1) {
        S = (int)((2 * cus) / (tiles128 > 0 ? tiles128 : 1));
        S = std::min(std::min(S, k / (8 * kSliceK)), 8);
      }
      const int sk = launch_reg_splitk(ctx, 128, S, g);
      return sk <= 0 ? sk : launch_reg(ctx, MMH_KERNEL_MFMA, g);
    }
    case MMH_KERNEL_MFMA_SPLITK_128X64: {   // K2s on 128x64 tiles
      int S = ctx ? ctx->splitk : 0;
      const long tiles = (long)((m + 127) / 128) * ((n + 63) / 64);
      if (S <= 1) {
        S = (int)((2 * cus) / (tiles > 0 ? tiles : 1));
        S = std::min(std::min(S, k / (8 * kSliceK)), 8);
      }
      const int sk = launch_reg_splitk(ctx, 64, S, g);
      return sk <= 0 ? sk : launch_reg(ctx, MMH_KERNEL_MFMA_128X64, g);
    }
#ifdef MMH_AB_BUILD
    case 45:
    case 46:
    case 47:
      return launch_dma(ctx, kernel, g);
    case 52: case 53: case 54: case 55: case 56: case 57: case 58: case 59:
      return launch_dma32(ctx, kernel, g);
    case 64: case 65: case 66: case 67: case 68: case 69: case 72: case 79: case 80: case 82: case 83: case 84: case 85: case 95: case 96: case 97: case 98: case 99:
      return launch_dma5(ctx, kernel, g);
    case 87: case 89: case 91: case 92: case 93: case 94:
      return launch_valu(ctx, kernel, g);
#endif
    default:
      return launch_reg(ctx, kernel, g);
  }
}

// What MMH_KERNEL_AUTO would do with a shape, as host arithmetic (mmh_auto_plan: no device, no launch): the tile it
// picks and, for the tiles whose residency the LDS alone decides, whether the launch would be the persistent
// stream-K form.  Uses the very functions the launch path uses (auto_kernel, streamk_wanted) on a default handle.
int auto_plan(int m, int n, int k, int lda, int ldb, int ldc, int base_align, int cu_count, int *kernel, long *tiles,
              int *streamk_grid) {
  if (m <= 0 || n <= 0 || k <= 0 || lda < k || ldb < n || ldc < n) return MMH_ERR_INVALID_ARG;
  mmh_context ctx;
  ctx.cu_count = cu_count > 0 ? cu_count : 256;
  // addresses that are never dereferenced: a 16-byte (or only 4-byte) aligned base for each operand
  const uintptr_t base = (uintptr_t)1 << 32, off = base_align >= 16 ? 0 : 4;
  const GemmArgs g{m, n, k, reinterpret_cast<const float *>(base + off), lda,
                   reinterpret_cast<const float *>(2 * base + off), ldb, reinterpret_cast<float *>(3 * base + off), ldc, 0, nullptr};
  const Plan plan = auto_plan_for(&ctx, g);
  const int kern = plan.kernel >= 0 ? plan.kernel : fallback_kernel(&ctx, g);
  int bm = plan.bm, bn = plan.bn;
  if (plan.kernel < 0) {
    switch (kern) {   // the register-staged fall-back tiles
      case MMH_KERNEL_MFMA_256X256: bm = 256; bn = 256; break;
      case MMH_KERNEL_MFMA: bm = 128; bn = 128; break;
      case MMH_KERNEL_MFMA_128X64: bm = 128; bn = 64; break;
      case MMH_KERNEL_MFMA_64X64: bm = 64; bn = 64; break;
      default: break;
    }
  }
  if (kernel) *kernel = kern;
  const long t = bm ? (long)((m + bm - 1) / bm) * ((n + bn - 1) / bn) : 0;
  if (tiles) *tiles = t;
  if (streamk_grid) {
    *streamk_grid = -1;
    if (plan.kernel >= 0) {
      *streamk_grid = 0;
      if (plan.form == 2) {   // (the grid the price was taken for: plan.sk_w workgroups per CU, GemmArgs::sk_w)
        const int grid = mmh::streamk_grid(t, ctx.cu_count, plan.sk_w);
        *streamk_grid = (grid > 0 && t % grid != 0) ? grid : 0;   // (launch_streamk: a count the grid divides runs plain)
      }
    }
  }
  return MMH_OK;
}

}  // namespace mmh
