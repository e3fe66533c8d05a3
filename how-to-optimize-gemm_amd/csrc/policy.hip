// policy.hip -- sgemm_on(): argument checks, the empty contraction, and MMH_KERNEL_AUTO's tile choice (the
// reference's `NEW := MMult_xxx` makefile switch, cuda/makefile:1-3, as a run-time decision).  Pure host code;
// the kernels are launched by launch_reg.hip / launch_dma.hip / launch_valu.hip.
#include <algorithm>

#include "internal.hpp"
#include "launch_common.hpp"   // streamk_wanted

namespace mmh {

namespace {
constexpr int kSliceK = 32;   // K-slice depth of the MFMA tiles (sgemm_tile.hpp BK)

// MMH_KERNEL_AUTO: tile choice by how well the shape fills 256 CUs (measured: profiles/r01_sweep.md,
// r02_ablation.md section 4, r03_offgrid_vs_vendor.md).  Tile counts are counted with the edge tiles a ragged
// shape needs; since round 3 the LDS-DMA tiles take ragged and 4-byte-aligned shapes too (guarded
// instantiations), so the same rules serve shapes on and off the 128-grid.
int auto_kernel(mmh_context *ctx, const GemmArgs &g) {
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const int m = g.m, n = g.n, k = g.k;
  const long tiles128 = (long)((m + 127) / 128) * ((n + 127) / 128);
  const long tiles128x64 = (long)((m + 127) / 128) * ((n + 63) / 64);
  const long tiles256 = (long)((m + 255) / 256) * ((n + 255) / 256);
  const long tiles64 = (long)((m + 63) / 64) * ((n + 63) / 64);
  const bool dma64 = dma_shape_ok(ctx, MMH_KERNEL_MFMA_64X64_DMA, g);
  const bool dma128x64 = dma_shape_ok(ctx, MMH_KERNEL_MFMA_128X64_DMA, g);
  const bool dma128 = dma_shape_ok(ctx, MMH_KERNEL_MFMA_128X128_DMA, g);
  // how much of the tiles' area is matrix (an edge tile costs a whole tile's time)
  auto fill = [&](long tiles, double area) { return (double)m * (double)n / ((double)tiles * area); };
  // Whole rounds of the 128x64 tile (two workgroups per CU) on problems with K loops long enough to amortise its
  // larger prologue: N = 4096 (2048 tiles = four rounds), 6144, 8192, 4096 x 8192 x 4096.  Measured level with the
  // 64x64 tile there (4096: 150.5 vs 150.3; 6144: 151.4 vs 151.5; 8192: 152.0 vs 152.3, tools/tile_ab.py) with 20 % less
  // fabric traffic (1.38 vs 1.65 GB per launch at 4096), a faster start from an idle clock (133 vs 130 TFLOP/s over a
  // process's launches 2 .. 21) -- and on the one box of four that ran the 64x64 tile 3 % slow at every many-tile size
  // (profiles/r03_notes.md section 7) the 128x64 sizes of the sweep lost 0-1 %: the smaller the tile, the more a launch
  // leans on the fabric.  (K = 1024: 145.1 vs 146.1 for the small tile, which keeps those.)
  if (dma128x64 && k >= 2048 && k <= 8192 && tiles128x64 >= 4 * cus && tiles128x64 % (2 * cus) == 0 &&
      tiles128x64 <= 32 * cus && fill(tiles128x64, 8192.0) >= 0.97)
    return MMH_KERNEL_MFMA_128X64_DMA;
  // The 64x64 LDS-DMA tile with three workgroups co-resident per CU shares the most efficient loop of all with the
  // 128x64 tile (148.5-150 TFLOP/s at N = 3072, where 2304 tiles are exactly nine per CU; 151.6-152.9 at 5120 .. 8192
  // against 148.5-150.4 for the 256x256 tile) -- on a PLAIN launch: under the chained stream-K launch its workgroups
  // run at different K phases and stop sharing operand slices in L2 (hit rate 81 % -> 22 %, 2.4 GB of fabric traffic
  // per launch, profiles/r02_ablation.md section 9).  So it is chosen for shapes with many tiles (>= 5.9 per CU) that
  // fill their last round of CUs to >= 97.5 % and that the rule above did not take (N = 2688, 3072, 3200 on the
  // reference sweep; 5120; the short-K shapes).
  // (Round 2 kept N = 4096 on the 256x256 tile because that tile follows the chip's clock ramp faster -- 137-139
  // TFLOP/s over the first 20 launches from an idle clock against 130-134 for the small tiles,
  // profiles/r03_cold_start.txt.  With launch #1 of a process no longer carrying 2 ms of one-offs (mmh_create warms
  // the handle) every tile clears the 80 % target under the reference's no-warm-up convention, and the sustained
  // rate -- what the headline metric quotes -- is the small tiles' by 1.5 %.)
  {
    const long rounds64 = (tiles64 + cus - 1) / cus;
    // ... and only up to N = 8192-sized problems: with K = 16384 and B beyond the Infinity Cache the
    // small tile's two slices of look-ahead no longer cover its misses (2048 .. 16384 x 16384 x 16384:
    // 147.8 .. 140.0 against 150.9-151.1 for the 256x256 tile, which those shapes keep)
    // (Ragged shapes: what counts is padding no worse than the alternative's -- one element past a 64-boundary pads
    // a 64x64 grid by 4-5 % and a 128x64 grid by 6-7 %: N = 2817, 3329, 3457, 3585 read 137.5 / 141.3 / 140.2 / 140.4 here
    // against 133.1 / 135.8 / 137.4 / 138.1 under the 128x64 stream-K launch, profiles/r03_offgrid_vs_vendor.md; 39 x 39
    // tiles at N = 2433 are 5.94 per CU: 134.3 against 126.8.)
    if (dma64 && k <= 8192 && tiles64 <= 64 * cus && tiles64 * 100 >= 590 * cus &&
        tiles64 * 1000 >= rounds64 * cus * 975 &&
        (fill(tiles64, 4096.0) >= 0.97 || fill(tiles64, 4096.0) >= fill(tiles128x64, 8192.0) + 0.015))
      return MMH_KERNEL_MFMA_64X64_DMA;
  }
  // a ragged count of 256x256 tiles would run as stream-K with ~1.1-1.2 tiles per workgroup; the 128x64
  // tile covers the same shape with >= 9 tiles per workgroup pair, phase-ordered (N = 4352 / 4608:
  // 148.6 / 148.9 against 147.2 / 147.4)
  if (dma128x64 && tiles256 >= cus && tiles256 % cus != 0 && k <= 8192 && tiles128x64 <= 32 * cus &&
      tiles128x64 * 10 >= 2 * cus * 18)
    return MMH_KERNEL_MFMA_128X64_DMA;
  // At least one 256x256 tile per CU: the big tile (fewest staging ops per MFMA) -- unless its
  // edge tiles pad the shape noticeably more than 128x128 tiles would (ragged tile COUNTS are balanced
  // by stream-K for either size).
  if (tiles256 >= cus && fill(tiles256, 65536.0) >= fill(tiles128, 16384.0) - 0.015) return MMH_KERNEL_MFMA_256X256;
  // Below one 256x256 tile per CU (N < 4096 on the reference sweep) the LDS-DMA tiles (sgemm_dma.hpp).  Two
  // co-resident 128x64 workgroups per CU under a phase-ordered stream-K launch (>= 1.8 tiles per workgroup of the
  // 2-per-CU grid: N >= 2816) are the best form there is for these sizes (145-147 TFLOP/s, 1-1.5 % ahead of one
  // 128x128 workgroup per CU).
  if (dma128x64 && tiles128x64 * 10 >= 2 * cus * 18) return MMH_KERNEL_MFMA_128X64_DMA;
  // In between, the candidates are scored: (share of the tiles' area that is matrix -- an edge tile costs a whole
  // tile's time) x (what the tile's loop sustains in the launch form it would get), the latter measured on the
  // square sweep and on the off-grid sweep (profiles/r03_offgrid_vs_vendor.md; TFLOP/s):
  //   128x128, stream-K or plain, one workgroup per CU, >= 1 tile per CU ................ 139
  //   128x64, two workgroups per CU (>= 2 tiles per CU) .................................. 138 (whole rounds: 142)
  //   128x64, ONE workgroup per CU (1 .. 2 tiles per CU: nobody to hide its stalls) ...... 124
  //   64x64 with two workgroups per CU (2 .. 3 tiles per CU, stream-K or plain) .......... 130
  // (64x64 with three workgroups per CU under a plain-order stream-K launch is erratic -- 97 .. 133 between
  // N = 1800 and 2200, its ranges start at unrelated K phases and thrash L2 -- and is not a candidate here; with
  // about one tile per CU it sustains ~104-117, which only the smallest shapes, below, settle for.)
  // This is what keeps a shape one element past a tile boundary of the big tiles (N = 2049, 2177, 2433, 2561: 9-11 %
  // of a 128x128 grid would be padding) on the tile that pads it least.
  {
    int best = -1;
    double best_score = 0.0;
    auto consider = [&](int kernel, bool ok, double score) {
      if (ok && score > best_score) { best_score = score; best = kernel; }
    };
    consider(MMH_KERNEL_MFMA_128X128_DMA, dma128 && tiles128 >= cus, fill(tiles128, 16384.0) * 139.0);
    consider(MMH_KERNEL_MFMA_128X64_DMA, dma128x64 && tiles128x64 * 100 >= cus * 125,
             fill(tiles128x64, 8192.0) * (tiles128x64 >= 2 * cus ? (tiles128x64 % (2 * cus) == 0 ? 142.0 : 138.0) : 124.0));
    consider(MMH_KERNEL_MFMA_64X64_DMA, dma64 && tiles64 >= 2 * cus && tiles64 < 3 * cus, fill(tiles64, 4096.0) * 130.0);
    if (best >= 0) return best;
  }
  if (dma64) return MMH_KERNEL_MFMA_64X64_DMA;
  // operands the descriptors cannot window (beyond 2 GiB), or the guarded LDS-DMA form switched off:
  // the register-staged tiles
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
    // MMH_OPT_RIM (off by default): a few rows / columns past a 64-boundary (N = 1025): the tiles take the trimmed shape,
    // the rim runs on the vector ALU in the same launch (sgemm_dma.hpp, "the rim") -- where the trimmed shape is a
    // one-round plain launch of the 64-wide LDS-DMA tiles; otherwise the whole shape goes the usual way, edge tiles and all.
    if (ctx && ctx->rim > 0) {
      const int rm = m % 64, rn = n % 64;
      if ((rm || rn) && rm <= ctx->rim && rn <= ctx->rim && m - rm >= 256 && n - rn >= 256 && k >= 64) {
        GemmArgs t = g;
        t.m = m - rm;
        t.n = n - rn;
        t.rim_m = m;
        t.rim_n = n;
        const int k0 = auto_kernel(ctx, t);
        if (k0 == MMH_KERNEL_MFMA_64X64_DMA || k0 == MMH_KERNEL_MFMA_128X64_DMA) {
          const int d = launch_dma(ctx, k0, t);
          if (d <= 0) return d;
        }
      }
    }
    kernel = auto_kernel(ctx, g);
  }
  switch (kernel) {
    case MMH_KERNEL_VALU:
    case MMH_KERNEL_VALU_128X128:
    case MMH_KERNEL_VALU_64X64:
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
    case MMH_KERNEL_MFMA_96X96_DMA5: {
      const int d = launch_dma5(ctx, kernel, g);
      return d <= 0 ? d : launch_reg(ctx, MMH_KERNEL_MFMA, g);
    }
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
    case 64: case 65: case 66: case 67: case 68: case 69: case 70: case 71: case 72: case 73: case 74: case 75: case 76: case 77:
    case 78: case 79: case 80:
      return launch_dma5(ctx, kernel, g);
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
  const int kern = auto_kernel(&ctx, g);
  int bm = 0, bn = 0, per_cu = 0;
  switch (kern) {
    case MMH_KERNEL_MFMA_64X64_DMA: bm = 64; bn = 64; per_cu = 3; break;     // 48 KiB ring
    case MMH_KERNEL_MFMA_128X64_DMA: bm = 128; bn = 64; per_cu = 2; break;   // 72 KiB
    case MMH_KERNEL_MFMA_128X128_DMA: bm = 128; bn = 128; per_cu = 1; break; // 96 KiB
    case MMH_KERNEL_MFMA_256X256: bm = 256; bn = 256; per_cu = 1; break;     // 128 KiB
    case MMH_KERNEL_MFMA: bm = 128; bn = 128; break;
    case MMH_KERNEL_MFMA_128X64: bm = 128; bn = 64; break;
    case MMH_KERNEL_MFMA_64X64: bm = 64; bn = 64; break;
    default: break;
  }
  if (kernel) *kernel = kern;
  const long t = bm ? (long)((m + bm - 1) / bm) * ((n + bn - 1) / bn) : 0;
  if (tiles) *tiles = t;
  if (streamk_grid) *streamk_grid = per_cu ? streamk_wanted(&ctx, t, bm, bn, per_cu) : -1;
  return MMH_OK;
}

}  // namespace mmh
