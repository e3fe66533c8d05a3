// launch_common.hpp -- what the kernel-launching translation units (launch_reg.hip, launch_dma.hip) share:
// the > 64 KiB LDS opt-in, per-kernel residency, and the persistent stream-K launch (sgemm_mfma.hpp, K2p).
#pragma once
#include <algorithm>

#include "internal.hpp"
#include "sgemm_tile.hpp"

namespace mmh {

constexpr size_t lds_bytes(int BM, int BN, int KB = BK) { return 2ull * (size_t)KB * (BM + BN) * sizeof(float); }

template <typename K>
int allow_big_lds(K kernel, size_t bytes) {
  HIP_TRY(opt_in_big_lds(reinterpret_cast<const void *>(kernel), bytes));
  return MMH_OK;
}

// resident workgroups per CU of a persistent kernel: what the runtime reports, never more than the
// LDS allows; computed once per handle (= per device) and kernel
template <typename K>
int resident_per_cu(mmh_context *ctx, K kernel, int threads, size_t lds) {
  const void *key = reinterpret_cast<const void *>(kernel);
  for (const auto &e : ctx->per_cu)
    if (e.first == key) return e.second;
  int v = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, threads, lds) != hipSuccess || v < 1) v = 1;
  const int by_lds = (int)((160 * 1024) / lds);
  v = v < by_lds ? v : by_lds;
  if (v < 1) v = 1;
  ctx->per_cu.emplace_back(key, v);
  return v;
}

// The grid of a persistent stream-K launch for `tiles` tiles: the largest whole number w <= per_cu of
// workgroups per CU that still gives every workgroup at least one full tile (the kernel's two-parts-per-
// tile hand-over assumes it; shorter ranges also measured slower than one workgroup per CU, N=2048: 75 vs
// 123 TFLOP/s).  0 = too few tiles.
inline int streamk_grid(long tiles, int cus, int per_cu) {
  for (int w = per_cu; w >= 1; --w)
    if (tiles >= (long)w * cus) return w * cus;
  return 0;
}

// Whether (and on how many persistent workgroups) a count of BM x BN tiles runs as a stream-K launch: 0 = the
// plain one-workgroup-per-tile launch serves it.
inline int streamk_wanted(const mmh_context *ctx, long tiles, int BM, int BN, int per_cu) {
  const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  const int grid = streamk_grid(tiles, cus, per_cu);
  if (grid == 0) return 0;                         // too few tiles
  if (tiles % grid == 0)                           // already balanced: persistent only on request, from two tiles per workgroup
    return ctx->persist && tiles >= 2L * grid && tiles <= (1L << 24) ? grid : 0;
  if (tiles > (1L << 24)) return 0;
  // Small tiles in nearly full rounds: the plain launch idles less than the hand-overs cost (measured,
  // N = 1408 on 64x64 tiles: 484 tiles for 512 slots run 119 TFLOP/s plain, 109 under stream-K).  From
  // 128x128 tiles up a hand-over is small beside a tile's work and stream-K wins whenever the count is
  // ragged (N = 2816 / 3456 / 3968: 143 / 145 / 146.5 against 138 / 139 / 138 plain).
  // (Balance is a matter of CUs, not of workgroup slots: co-resident workgroups share their CU's matrix pipe.)
  // (The 128x64 tile counts as a big one once the launch is phase-ordered -- >= 1.8 tiles per workgroup,
  // sk_tables_for -- N = 3968: 147.7 under stream-K, 141.9 plain.)
  const bool ordered = ctx->sk_order && tiles * 10 >= (long)grid * ctx->sk_order_min10;
  if (ctx->streamk != 2 && BM * BN < 128 * 128 && !(ordered && BM * BN >= 128 * 64)) {   // MMH_OPT_STREAMK = 2: whenever ragged
    const long rounds = (tiles + cus - 1) / cus;
    if (tiles * 100 >= rounds * cus * 93) return 0;
  }
  return grid;
}

// Persistent chained stream-K launch: what is common to every tile code.  `kern` is the instantiation
// to launch (`occ_kern` the one whose residency bounds the grid).  Returns MMH_OK if it launched, 1 if
// the shape does not qualify (caller then uses the plain one-tile-per-workgroup launch).  `force`: launch
// even when the policy below prefers the plain launch (warm-up, tools).
template <typename K>
int launch_streamk(mmh_context *ctx, K kern, K occ_kern, int BM, int BN, int KB, int threads, size_t lds, const char *what,
                   const GemmArgs &g, long decide_tiles = 0, int order_min10 = 0) {
  const int nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
  const long tiles = (long)nbm * nbn;
  const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  {
    const int ok = allow_big_lds(kern, lds);
    if (ok != MMH_OK) return ok;
  }
  int per_cu = resident_per_cu(ctx, occ_kern, threads, lds);
  // MMH_KERNEL_AUTO priced a grid of g.sk_w workgroups per CU (the family's stream-K residency in policy_table.inc, which
  // tests/test_kernel_resources.py holds to what the binary's registers allow): what is priced is what is launched
  if (g.sk_w > 0 && g.sk_w < per_cu) per_cu = g.sk_w;
  // g.form (MMH_KERNEL_AUTO's cost table has decided): 1 = plain, 2 = persistent whenever the count is ragged; 0 = a
  // kernel the caller forced: the rule of streamk_wanted.  (decide_tiles: the count that rule looks at, when it is
  // not the launch's own -- thin edge tiles of the K2W kernels; a persistent launch then still covers all `tiles`.)
  if (g.form == 1) return 1;
  int grid = 0;
  if (g.form == 2) {
    grid = streamk_grid(tiles, cus, per_cu);
    if (grid == 0 || tiles % grid == 0 || tiles > (1L << 24)) return 1;
  } else {
    // (MMH_OPT_STREAMK = 2 -- "whenever the count is ragged", the datasets' /sk2 -- looks at the launch's own count only)
    if (decide_tiles > 0 && ctx->streamk != 2 && streamk_wanted(ctx, decide_tiles, BM, BN, per_cu) == 0) return 1;
    grid = streamk_wanted(ctx, tiles, BM, BN, per_cu);
    if (grid == 0) return 1;
  }
  if (tiles * (long)((g.k + KB - 1) / KB) >= (1L << 31)) return 1;   // the kernels split tiles x K-slices in 32-bit arithmetic
  int *flags = nullptr;
  float *parts = nullptr;   // one partial-tile slot per range
  int rc = workspace_for(ctx, g.s, tiles, (size_t)grid * BM * BN * sizeof(float), &flags, &parts);
  if (rc != MMH_OK) return rc;
  // The ranges assume every workgroup owns 1/w of a CU.  When more than w would FIT (a 48 KiB ring three
  // times), nothing obliges the dispatcher to spread grid = w x CUs workgroups evenly -- seen as a bimodal
  // rate (N = 1536: 130 or 100 TFLOP/s from run to run) -- so the launch asks for 160 KiB / w of LDS:
  // exactly w workgroups fit, every CU gets its share.  (Speed only: the hand-over is wait-free.)
  size_t lds_launch = lds;
  if (ctx->pin) {
    const size_t share = ((size_t)(160 * 1024) / (size_t)(grid / cus)) & ~(size_t)255;
    if (share > lds_launch) lds_launch = share;
    const int ok = allow_big_lds(kern, lds_launch);
    if (ok != MMH_OK) return ok;
  }
  const int *order = nullptr, *place = nullptr;
  if ((rc = sk_tables_for(ctx, tiles, (g.k + KB - 1) / KB, grid, g.s, &order, &place, order_min10)) != MMH_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds_launch, g.s, g.m, g.n, g.k, g.A, g.lda, g.B, g.ldb, g.C,
                     g.ldc, g.acc, nbm, nbn, flags, parts, order, place, ctx->sk_stats);
  HIP_TRY(hipGetLastError());
  {
    char buf[224];
    snprintf(buf, sizeof buf, "%s, %ld tiles on %d persistent workgroups%s", what, tiles, grid,
             order ? ", phase-ordered" : "");
    set_last_launch(buf);
  }
  return MMH_OK;
}

// One-tile warm-up of a stream-K instantiation: grid 1, one whole tile (no hand-over, no workspace).
template <typename K>
int warm_streamk_kernel(K kern, int BM, int BN, int KB, int threads, size_t lds, float *scratch, hipStream_t s) {
  const int ok = allow_big_lds(kern, lds);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, s, BM, BN, KB, scratch, KB, scratch, BN, scratch + 65536, BN, 0, 1, 1,
                     static_cast<int *>(nullptr), static_cast<float *>(nullptr), static_cast<const int *>(nullptr),
                     static_cast<const int *>(nullptr), static_cast<int *>(nullptr));
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

}  // namespace mmh
