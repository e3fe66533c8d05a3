// mmult_hip.hip -- libmmult_hip.so: the C ABI declared in include/mmult_hip.h
// over the hand-written gfx950 kernels in this directory.
//
// This file is the "thin C-ABI shim" of BASELINE.json's north star: argument
// validation, kernel selection (the reference's `NEW := MMult_xxx` makefile
// switch, cuda/makefile:1-3, made a run-time choice), launch, and the
// host-pointer flavour's staging.  No torch, no CPU fallback: if no gfx950
// device is visible every compute entry point returns MMH_ERR_NO_DEVICE /
// MMH_ERR_HIP.
//
// Two builds of this one source:
//   libmmult_hip.so     the product: every kernel id it accepts returns correct results;
//   libmmult_hip_ab.so  (-DMMH_AB_BUILD, tools/ only) additionally carries the scheduling A/B
//                       variants and the TIMING-ONLY ablation builds whose results are wrong.
#include "../../include/mmult_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "igemm_s8.hpp"
#include "probes.hpp"
#include "quant_s8.hpp"
#include "sgemm_dma.hpp"
#include "sgemm_mfma.hpp"
#include "sgemm_valu.hpp"

namespace {

thread_local std::string g_last_error;
thread_local std::string g_last_launch;   // which kernel configuration the last sgemm call ran

int hip_fail(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return MMH_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return MMH_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
      g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
      return MMH_ERR_ALLOC;
    }
    bytes = need;
    return MMH_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// Entry points run on the HANDLE's device and leave the caller's current device as they found it
// (a torch process whose current device differs from the handle's must not find it changed).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev == device) return hipSuccess;
    e = hipSetDevice(device);
    switched = e == hipSuccess;
    return e;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

constexpr int kMaxHostPanels = 16;

}  // namespace

struct mmh_context {
  int device = 0;
  int kernel = MMH_KERNEL_AUTO;
  int cu_count = 0;
  DevBuf a, b, c;          // staging for the host-pointer flavour
  DevBuf bt;               // int8 GEMM: packed (transposed, padded) B
  int igemm_mode = 0;      // 0 auto (B in place / packed-B + LDS-DMA), 1 in-kernel transpose, 2 simple
  DevBuf qa, qb, qc, qs;   // quantised GEMM workspace: int8 A, int8 B, int32 C, {amax bits, scales}
  DevBuf flags;            // stream-K / split-K per-tile hand-off words; every launch leaves them ZERO (the
                           // last reader of a word resets it), so only a fresh or suspect buffer is memset
  bool flags_dirty = true;
  DevBuf parts;            // stream-K / split-K partial tiles
  int streamk = 1;         // allow the persistent stream-K launch for ragged tile counts
  int splitk = 0;          // opt-in split-K: 0 off (default), 1 auto, >= 2 that many parts
  int host_panels = -1;    // host flavour: -1 auto, 0/1 the plain staged form, n pipelined row panels
  void *rocblas = nullptr; // rocblas_handle, created on first use
  // the sticky error word: host memory the device can write (a hand-off wait that times out adds to it);
  // every entry point looks at it before doing anything else
  int *sticky = nullptr;       // host view
  int *sticky_dev = nullptr;   // device view of the same word
  long long spin_limit = 1ll << 26;
  int fault = 0;               // MMH_OPT_FAULT_INJECT
  int pin = 1;                 // persistent / sparse launches ask for 160 KiB / w of LDS so that exactly w workgroups fit a CU
  int sk_order = 1;            // stream-K launches get the phase-ordered range / tile tables (sk_tables below)
  // stream-K tables per launch shape (tiles, K-slices, grid): [order: grid ints][place: tiles ints]
  struct SkTable { long tiles = 0; int nk = 0, grid = 0; DevBuf buf; unsigned long stamp = 0; };
  std::vector<SkTable> sk_tables;
  unsigned long sk_stamp = 0;
  // the hand-off workspaces above are per handle: a launch on another stream first waits for the
  // stream that used them last
  hipStream_t ws_stream = nullptr;
  bool ws_used = false;
  // resident workgroups per CU of each persistent kernel, per handle (= per device)
  std::vector<std::pair<const void *, int>> per_cu;
  // host flavour pipeline: copy-in / compute / copy-out streams, per-panel events
  hipStream_t hs_in = nullptr, hs_run = nullptr, hs_out = nullptr;
  hipEvent_t ev_in[kMaxHostPanels] = {}, ev_run[kMaxHostPanels] = {}, ev_b = nullptr;
  bool pipeline_ready = false;
  hipEvent_t t0 = nullptr, t1 = nullptr;   // mmh_sgemm_host_timed
};

namespace {

// every compute entry point: refuse a handle whose sticky error word is set
int check_sticky(mmh_context *h) {
  if (h && h->sticky && *reinterpret_cast<volatile int *>(h->sticky) != 0) {
    g_last_error = "an earlier stream-K / split-K launch on this handle timed out waiting for a hand-off: "
                   "its result is invalid (clear with mmh_set_option(h, MMH_OPT_STREAMK_TIMEOUTS, 0))";
    return MMH_ERR_HIP;
  }
  return MMH_OK;
}
#define ENTER(h)                                       \
  DeviceGuard guard_;                                  \
  HIP_TRY(guard_.enter((h)->device));                  \
  if (int st_ = check_sticky(h); st_ != MMH_OK) return st_

constexpr size_t lds_bytes(int BM, int BN, int KB = mmh::BK) {
  return 2ull * (size_t)KB * (BM + BN) * sizeof(float);
}

template <typename K>
int allow_big_lds(K kernel, size_t bytes) {
  HIP_TRY(mmh::opt_in_big_lds(reinterpret_cast<const void *>(kernel), bytes));
  return MMH_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// whole tiles, 16-byte aligned operands: the unguarded instantiations
bool fast_shape(int BM, int BN, int KB, int m, int n, int k, const void *A, int lda, const void *B, int ldb,
                const void *C, int ldc) {
  return (m % BM == 0) && (n % BN == 0) && (k % KB == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) &&
         aligned16(A) && aligned16(B) && aligned16(C);
}

// the buffer-descriptor path needs every byte offset inside a 2 GiB window
bool window_ok(int BM, int BN, int k, int lda, int ldb) {
  const size_t lim = (1ull << 31) - 4096;
  return ((size_t)BM * lda + k) * 4 < lim && ((size_t)k * ldb + BN) * 4 < lim;
}

// SIMPLE: the un-pipelined rung.  SCHED / BUFLD / ABL: see sgemm_mfma.hpp.  The
// buffer-descriptor path needs every byte offset inside a 2 GiB window; larger
// operands fall back to 64-bit global addressing (same kernel, BUFLD = false).
template <int BM, int BN, bool SIMPLE = false, int SCHED = 4, int ABL = 0, bool BUFLD = true, int WTN = 4,
          int WTM = 4, int KB = mmh::BK>
int launch_mfma(int m, int n, int k, const float *A, int lda, const float *B, int ldb,
                float *C, int ldc, int acc, hipStream_t s) {
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  const bool fast = fast_shape(BM, BN, KB, m, n, k, A, lda, B, ldb, C, ldc);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  dim3 grid((unsigned)(nbm * nbn)), block(threads);
  const bool win = window_ok(BM, BN, k, lda, ldb);
#define MMH_LAUNCH(KERN)                                                                   \
  do {                                                                                     \
    auto kern = KERN;                                                                      \
    const int ok = allow_big_lds(kern, lds);                                               \
    if (ok != MMH_OK) return ok;                                                           \
    hipLaunchKernelGGL(kern, grid, block, lds, s, m, n, k, A, lda, B, ldb, C, ldc, acc, nbm, nbn); \
  } while (0)
  if constexpr (SIMPLE) {
    if (fast) MMH_LAUNCH((mmh::sgemm_mfma_simple_kernel<BM, BN, false>));
    else      MMH_LAUNCH((mmh::sgemm_mfma_simple_kernel<BM, BN, true>));
  } else if (!fast) {
    // guarded launch: buffer descriptors bound the reads (any alignment >= 4 B);
    // operands larger than the descriptor window use the per-element path
    if (BUFLD && win) MMH_LAUNCH((mmh::sgemm_mfma_kernel<BM, BN, true, SCHED, 0, true, WTN, WTM, KB>));
    else              MMH_LAUNCH((mmh::sgemm_mfma_kernel<BM, BN, true, SCHED, 0, false, WTN, WTM, KB>));
  } else if (BUFLD && win) {
    MMH_LAUNCH((mmh::sgemm_mfma_kernel<BM, BN, false, SCHED, ABL, BUFLD, WTN, WTM, KB>));
  } else {
    MMH_LAUNCH((mmh::sgemm_mfma_kernel<BM, BN, false, SCHED, ABL, false, WTN, WTM, KB>));
  }
#undef MMH_LAUNCH
  HIP_TRY(hipGetLastError());
  {
    char buf[160];
    snprintf(buf, sizeof buf, "%s<%d,%d> wave tile %dx%d, K-slice %d, %s%d workgroups of %d threads",
             SIMPLE ? "sgemm_mfma_simple_kernel" : "sgemm_mfma_kernel", BM, BN, 16 * WTM, 16 * WTN, KB,
             fast ? "" : "guarded, ", nbm * nbn, threads);
    g_last_launch = buf;
  }
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

// The hand-off workspaces (flags, partial tiles) belong to the handle.  Launches on ONE stream are
// ordered by the stream; an eager launch on ANOTHER stream than the last eager one first waits for
// that stream, so two streams can never have the workspaces in use at once.  A launch that is being
// CAPTURED into a hipGraph executes nothing now and may not synchronise anything: it is recorded as
// it is, and whoever replays the graph orders it against other work on the handle (as for any buffer
// a graph owns) -- include/mmult_hip.h says so.
int claim_workspaces(mmh_context *ctx, hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  if (cap != hipStreamCaptureStatusNone) return MMH_OK;
  if (ctx->ws_used && ctx->ws_stream != s) HIP_TRY(hipStreamSynchronize(ctx->ws_stream));
  ctx->ws_stream = s;
  ctx->ws_used = true;
  return MMH_OK;
}

// The hand-off words of `tiles` tiles, all zero.  The kernels restore the zeros themselves (the part that
// finishes a tile resets its counter), so the fill runs only when the buffer is new, has grown, or a
// launch may have died half-way (a sticky error was cleared).
int prepare_flags(mmh_context *ctx, long tiles, hipStream_t s, int **flags) {
  const size_t need = (size_t)tiles * sizeof(int);
  if (need > ctx->flags.bytes) {
    const int rc = ctx->flags.reserve(std::max(need, (size_t)(64u << 10)));
    if (rc != MMH_OK) return rc;
    ctx->flags_dirty = true;
  }
  if (ctx->flags_dirty) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    HIP_TRY(hipMemsetAsync(ctx->flags.p, 0, ctx->flags.bytes, s));
    // a fill recorded into a graph does not clean the buffer NOW: stay dirty until an eager launch
    if (cap == hipStreamCaptureStatusNone) ctx->flags_dirty = false;
  }
  *flags = static_cast<int *>(ctx->flags.p);
  return MMH_OK;
}

// The two tables of a stream-K launch (streamk_body's `order` and `place`), per shape, cached in the handle.
// Range r of the G ranges covers units [U r / G, U (r + 1) / G) of the U = tiles x nk (tile slot, K-slice)
// units; its PHASE is the length of its head (the slices of its last slot it computes first): its whole
// tiles start that many slice-times into the launch.
//   order[rho]: the range taken by the workgroup at chip position rho (XCD-contiguous) -- ranges sorted by
//               phase, so that neighbours on the chip are a slice or two apart in K, not half a tile;
//   place[j]  : the tile computed in slot j -- dealt out level by level (the o-th slot each range owns),
//               within a level in phase order: what neighbouring workgroups compute at the same time are
//               neighbouring tiles of the grouped raster.
// Any pair of bijections is CORRECT (the chain only needs every workgroup to agree on them); these restore
// the L2 reuse a plain launch has.  Built on the host at a shape's first eager launch (one synchronising
// copy); a launch that is being captured into a hipGraph runs with the identity tables.
// (pure host arithmetic: mmh_streamk_plan exposes it to the CPU tests)
bool build_sk_tables(long tiles, int nk, int grid, int *order, int *place) {
  if (tiles <= 0 || nk <= 0 || grid <= 0 || tiles < grid) return false;
  const long long U = (long long)tiles * nk;
  auto S = [&](long long r) { return U * r / grid; };
  std::vector<int> first(grid + 1);
  std::vector<std::pair<int, int>> by_phase(grid);
  for (int r = 0; r <= grid; ++r) first[r] = r == grid ? (int)tiles : (int)((S(r) + nk - 1) / nk);
  for (int r = 0; r < grid; ++r) by_phase[r] = {(int)(S(r + 1) % nk), r};
  std::sort(by_phase.begin(), by_phase.end());
  int levels = 0;
  for (int i = 0; i < grid; ++i) {
    order[i] = by_phase[i].second;
    levels = std::max(levels, first[by_phase[i].second + 1] - first[by_phase[i].second]);
  }
  int next = 0;
  for (int o = 0; o < levels; ++o)
    for (int i = 0; i < grid; ++i) {
      const int r = by_phase[i].second;
      if (first[r + 1] - first[r] > o) place[first[r] + o] = next++;
    }
  return next == (int)tiles;
}

int sk_tables_for(mmh_context *ctx, long tiles, int nk, int grid, hipStream_t s, const int **order, const int **place) {
  *order = *place = nullptr;
  // Worth it from ~1.8 tiles per workgroup (measured): phase order puts the two workgroups that share a
  // tile on different XCDs, so the partial tile crosses the fabric instead of being an L2 hit -- with one
  // tile per workgroup that hand-over is a tenth of the launch (N = 2176 on 128x64 tiles: 139.5 -> 125.0),
  // with two or more the restored L2 reuse wins (N = 3584 on 64x64 tiles: 138.5 -> 146.0).
  if (!ctx->sk_order || tiles > (1L << 20) || tiles * 10 < (long)grid * 18) return MMH_OK;
  // a launch that is being captured must not point into this cache (an entry can be evicted and rewritten
  // long before the graph is replayed): it is recorded with the identity tables
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  if (cap != hipStreamCaptureStatusNone) return MMH_OK;
  for (auto &t : ctx->sk_tables)
    if (t.tiles == tiles && t.nk == nk && t.grid == grid) {
      t.stamp = ++ctx->sk_stamp;
      *order = static_cast<const int *>(t.buf.p);
      *place = *order + grid;
      return MMH_OK;
    }
  std::vector<int> host((size_t)grid + (size_t)tiles);
  if (!build_sk_tables(tiles, nk, grid, host.data(), host.data() + grid)) return MMH_OK;   // identity is always right
  // a new shape: evict the least recently used entry beyond eight
  mmh_context::SkTable *slot = nullptr;
  if (ctx->sk_tables.size() < 8) {
    ctx->sk_tables.emplace_back();
    slot = &ctx->sk_tables.back();
  } else {
    slot = &ctx->sk_tables[0];
    for (auto &t : ctx->sk_tables)
      if (t.stamp < slot->stamp) slot = &t;
  }
  HIP_TRY(hipStreamSynchronize(s));              // the evicted tables may still be read by a launch on s
  if (ctx->ws_used && ctx->ws_stream != s) HIP_TRY(hipStreamSynchronize(ctx->ws_stream));
  const int rc = slot->buf.reserve(host.size() * sizeof(int));
  if (rc != MMH_OK) { slot->tiles = 0; return rc; }
  HIP_TRY(hipMemcpy(slot->buf.p, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
  slot->tiles = tiles; slot->nk = nk; slot->grid = grid; slot->stamp = ++ctx->sk_stamp;
  *order = static_cast<const int *>(slot->buf.p);
  *place = *order + grid;
  return MMH_OK;
}

// Persistent chained stream-K launch (sgemm_mfma.hpp, K2p): what is common to every tile code.  `kern`
// is the instantiation to launch (`occ_kern` the one whose residency bounds the grid).  Returns MMH_OK
// if it launched, 1 if the shape does not qualify (caller then uses the plain one-tile-per-workgroup
// launch).
template <typename K>
int launch_streamk(mmh_context *ctx, K kern, K occ_kern, int BM, int BN, int KB, int threads, size_t lds, const char *what,
                   int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int acc,
                   hipStream_t s) {
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  const long tiles = (long)nbm * nbn;
  const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  {
    const int ok = allow_big_lds(kern, lds);
    if (ok != MMH_OK) return ok;
  }
  const int per_cu = resident_per_cu(ctx, occ_kern, threads, lds);
  // the largest grid (whole CUs' worth of workgroups) that still gives every
  // workgroup at least one full tile, so that chains never stall.  (Shorter ranges
  // are legal for the kernel -- tiles then have three or more parts -- but measured
  // slower than one workgroup per CU: the parts of a tile run one after the other
  // whoever computes them, N=2048: 75 vs 123 TFLOP/s.)
  int grid = 0;
  for (int w = per_cu; w >= 1; --w)
    if (tiles >= (long)w * cus) { grid = w * cus; break; }
  if (grid == 0 || tiles % grid == 0) return 1;   // too few tiles, or already balanced
  if (tiles > (1L << 24)) return 1;
  // Small tiles in nearly full rounds: the plain launch idles less than the hand-overs cost (measured,
  // N = 1408 on 64x64 tiles: 484 tiles for 512 slots run 119 TFLOP/s plain, 109 under stream-K).  From
  // 128x128 tiles up a hand-over is small beside a tile's work and stream-K wins whenever the count is
  // ragged (N = 2816 / 3456 / 3968: 143 / 145 / 146.5 against 138 / 139 / 138 plain).
  // (Balance is a matter of CUs, not of workgroup slots: co-resident workgroups share their CU's matrix pipe.)
  // (The 128x64 tile counts as a big one once the launch is phase-ordered -- >= 1.8 tiles per workgroup,
  // sk_tables_for -- N = 3968: 147.7 under stream-K, 141.9 plain.)
  const bool ordered = ctx->sk_order && tiles * 10 >= (long)grid * 18;
  if (ctx->streamk != 2 && BM * BN < 128 * 128 && !(ordered && BM * BN >= 128 * 64)) {   // MMH_OPT_STREAMK = 2: whenever ragged
    const long rounds = (tiles + cus - 1) / cus;
    if (tiles * 100 >= rounds * cus * 93) return 1;
  }
  int rc = claim_workspaces(ctx, s);
  if (rc != MMH_OK) return rc;
  int *flags = nullptr;
  if ((rc = prepare_flags(ctx, tiles, s, &flags)) != MMH_OK) return rc;
  rc = ctx->parts.reserve((size_t)grid * BM * BN * sizeof(float));   // one partial-tile slot per range
  if (rc != MMH_OK) return rc;
  float *parts = static_cast<float *>(ctx->parts.p);
  // The ranges assume every workgroup owns 1/w of a CU.  When more than w would FIT (a 48 KiB ring three
  // times), nothing obliges the dispatcher to spread grid = w x CUs workgroups evenly -- seen as a bimodal
  // rate (N = 1536: 130 or 100 TFLOP/s from run to run) -- so the launch asks for 160 KiB / w of LDS:
  // exactly w workgroups fit, every CU gets its share.
  size_t lds_launch = lds;
  if (ctx->pin) {
    const size_t share = ((size_t)(160 * 1024) / (size_t)(grid / cus)) & ~(size_t)255;
    if (share > lds_launch) lds_launch = share;
    const int ok = allow_big_lds(kern, lds_launch);
    if (ok != MMH_OK) return ok;
  }
  const int *order = nullptr, *place = nullptr;
  if ((rc = sk_tables_for(ctx, tiles, (k + KB - 1) / KB, grid, s, &order, &place)) != MMH_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds_launch, s, m, n, k, A, lda, B, ldb, C, ldc, acc, nbm,
                     nbn, flags, ctx->sticky_dev, parts, ctx->spin_limit, ctx->fault, order, place);
  HIP_TRY(hipGetLastError());
  {
    char buf[200];
    snprintf(buf, sizeof buf, "%s, %ld tiles on %d persistent workgroups", what, tiles, grid);
    g_last_launch = buf;
  }
  return MMH_OK;
}

// ... of the register-staged tile config <BM, BN, WTN>
template <int BM, int BN, int WTN, int WTM = 4, int KB = mmh::BK>
int try_launch_streamk(mmh_context *ctx, int m, int n, int k, const float *A, int lda,
                       const float *B, int ldb, float *C, int ldc, int acc, hipStream_t s) {
  if (!ctx || !ctx->streamk || !ctx->sticky_dev) return 1;
  if (!window_ok(BM, BN, k, lda, ldb)) return 1;   // descriptor window
  // whole, 16-byte-aligned shapes run the unguarded kernel; everything else the guarded one (partial
  // tiles travel through a workspace, not through C, so C's alignment and ragged edges do not matter)
  const bool fast = fast_shape(BM, BN, KB, m, n, k, A, lda, B, ldb, C, ldc);
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  auto kern_fast = mmh::sgemm_mfma_streamk_kernel<BM, BN, false, WTN, WTM, KB>;
  auto kern_edge = mmh::sgemm_mfma_streamk_kernel<BM, BN, true, WTN, WTM, KB>;
  char what[160];
  snprintf(what, sizeof what, "sgemm_mfma_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d%s", BM, BN, 16 * WTM,
           16 * WTN, KB, fast ? "" : ", guarded");
  return launch_streamk(ctx, fast ? kern_fast : kern_edge, kern_edge, BM, BN, KB, threads, lds, what, m, n, k, A, lda, B, ldb,
                        C, ldc, acc, s);
}

// ... of the LDS-DMA tile (sgemm_dma.hpp): whole-tile, 16-byte aligned shapes only
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
int try_launch_streamk_dma(mmh_context *ctx, int m, int n, int k, const float *A, int lda, const float *B, int ldb,
                           float *C, int ldc, int acc, hipStream_t s) {
  using T = mmh::DmaTile<BM, BN, KB, WTM, WTN, NBUF>;
  if (!ctx || !ctx->streamk || !ctx->sticky_dev) return 1;
  if (!fast_shape(BM, BN, KB, m, n, k, A, lda, B, ldb, C, ldc) || !window_ok(BM, BN, k, lda, ldb)) return 1;
  auto kern = mmh::sgemm_dma_streamk_kernel<BM, BN, KB, WTM, WTN, NBUF>;
  char what[160];
  snprintf(what, sizeof what, "sgemm_dma_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by LDS-DMA", BM,
           BN, 16 * WTM, 16 * WTN, KB, NBUF);
  return launch_streamk(ctx, kern, kern, BM, BN, KB, T::THREADS, T::LDS_BYTES, what, m, n, k, A, lda, B, ldb, C, ldc, acc, s);
}

// Opt-in split-K launch (sgemm_mfma.hpp, K2s) of tile config <BM, BN, WTN> with S concurrent K parts.
// Returns MMH_OK if it launched, 1 if the shape does not qualify.
template <int BM, int BN, int WTN, int WTM = 4, int KB = mmh::BK>
int try_launch_splitk(mmh_context *ctx, int S, int m, int n, int k, const float *A, int lda, const float *B,
                      int ldb, float *C, int ldc, int acc, hipStream_t s) {
  if (!ctx || !ctx->sticky_dev || S < 2) return 1;
  if (!window_ok(BM, BN, k, lda, ldb) || !fast_shape(BM, BN, KB, m, n, k, A, lda, B, ldb, C, ldc)) return 1;
  const int nbm = m / BM, nbn = n / BN, nk = k / KB;
  const long tiles = (long)nbm * nbn;
  if (S > nk) S = nk;
  if (S < 2) return 1;
  const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  auto kern = mmh::sgemm_mfma_splitk_kernel<BM, BN, WTN, WTM, KB>;
  {
    const int ok = allow_big_lds(kern, lds);
    if (ok != MMH_OK) return ok;
  }
  const int per_cu = resident_per_cu(ctx, kern, threads, lds);
  while (S >= 2 && tiles * S > (long)per_cu * cus) --S;   // every part resident at once
  if (S < 2) return 1;
  int rc = claim_workspaces(ctx, s);
  if (rc != MMH_OK) return rc;
  int *flags = nullptr;
  if ((rc = prepare_flags(ctx, tiles, s, &flags)) != MMH_OK) return rc;
  rc = ctx->parts.reserve((size_t)tiles * (S - 1) * BM * BN * sizeof(float));
  if (rc != MMH_OK) return rc;
  float *parts = static_cast<float *>(ctx->parts.p);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * S)), dim3(threads), lds, s, m, n, k, A, lda, B, ldb, C, ldc, acc,
                     nbm, nbn, S, flags, ctx->sticky_dev, parts, ctx->spin_limit);
  HIP_TRY(hipGetLastError());
  {
    char buf[176];
    snprintf(buf, sizeof buf,
             "sgemm_mfma_splitk_kernel<%d,%d> wave tile %dx%d, K-slice %d, %ld tiles x %d concurrent K parts",
             BM, BN, 16 * WTM, 16 * WTN, KB, tiles, S);
    g_last_launch = buf;
  }
  return MMH_OK;
}

// K2L (sgemm_dma.hpp): both operands by LDS-DMA.  Whole-tile, 16-byte aligned shapes inside the descriptor
// window only; returns 1 when the shape does not qualify (the caller then runs the register-staged kernel).
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF>
int try_launch_dma(int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int acc,
                   hipStream_t s) {
  using T = mmh::DmaTile<BM, BN, KB, WTM, WTN, NBUF>;
  if (!fast_shape(BM, BN, KB, m, n, k, A, lda, B, ldb, C, ldc) || !window_ok(BM, BN, k, lda, ldb)) return 1;
  const int nbm = m / BM, nbn = n / BN;
  auto kern = mmh::sgemm_mfma_dma_kernel<BM, BN, KB, WTM, WTN, NBUF>;
  const int ok = allow_big_lds(kern, T::LDS_BYTES);
  if (ok != MMH_OK) return ok;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(T::THREADS), T::LDS_BYTES, s, m, n, k, A, lda, B, ldb, C,
                     ldc, acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  char buf[176];
  snprintf(buf, sizeof buf, "sgemm_mfma_dma_kernel<%d,%d> wave tile %dx%d, K-slice %d x %d ring buffers by LDS-DMA, %d workgroups of %d threads",
           BM, BN, 16 * WTM, 16 * WTN, KB, NBUF, nbm * nbn, T::THREADS);
  g_last_launch = buf;
  return MMH_OK;
}

template <int BM, int BN, int KB>
int launch_valu_tile(int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C,
                     int ldc, int acc, hipStream_t s) {
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  const bool fast = fast_shape(BM, BN, KB, m, n, k, A, lda, B, ldb, C, ldc);
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  dim3 grid((unsigned)(nbm * nbn)), block(256);
  if (fast)
    hipLaunchKernelGGL((mmh::sgemm_valu_kernel<BM, BN, KB, false>), grid, block, lds, s, m, n, k, A, lda, B, ldb,
                       C, ldc, acc, nbm, nbn);
  else
    hipLaunchKernelGGL((mmh::sgemm_valu_kernel<BM, BN, KB, true>), grid, block, lds, s, m, n, k, A, lda, B, ldb,
                       C, ldc, acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  char buf[96];
  snprintf(buf, sizeof buf, "sgemm_valu_kernel<%d,%d> K-slice %d, %s%d workgroups", BM, BN, KB, fast ? "" : "guarded, ",
           nbm * nbn);
  g_last_launch = buf;
  return MMH_OK;
}

int launch_naive(int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C,
                 int ldc, int acc, hipStream_t s) {
  dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 3) / 4)), block(256);
  hipLaunchKernelGGL(mmh::sgemm_naive_kernel, grid, block, 0, s, m, n, k, A, lda, B, ldb, C, ldc,
                     acc);
  HIP_TRY(hipGetLastError());
  g_last_launch = "sgemm_naive_kernel";
  return MMH_OK;
}

int check_gemm_args(int m, int n, int k, const void *A, int lda, const void *B, int ldb,
                    const void *C, int ldc) {
  if (m < 0 || n < 0 || k < 0) return MMH_ERR_INVALID_ARG;
  if (m == 0 || n == 0) return MMH_OK;
  if (!C || ldc < n) return MMH_ERR_INVALID_ARG;
  if (k > 0 && (!A || !B || lda < k || ldb < n)) return MMH_ERR_INVALID_ARG;
  return MMH_OK;
}

// is `kernel` an id this build accepts?
bool known_kernel(int kernel) { return mmh_kernel_name(kernel) != nullptr; }

int sgemm_on(mmh_context *ctx, int kernel, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
             float *dC, int ldc, int accumulate, hipStream_t s) {
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) {
    g_last_error = "invalid argument";
    return rc;
  }
  if (m == 0 || n == 0) return MMH_OK;
  if (k == 0) {
    // empty contraction: C = 0 (overwrite) or C unchanged (accumulate)
    if (!accumulate)
      HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float),
                               (size_t)m, s));
    return MMH_OK;
  }
  const int acc = accumulate ? 1 : 0;
  const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
  const long tiles128 = (long)((m + 127) / 128) * ((n + 127) / 128);
  switch (kernel) {
    case MMH_KERNEL_VALU:
      // K1: the 128x128 rung from half a tile per CU up, the 64x64 tile below (measured, N = 1024 ..
      // 2048: 33 / 54 TFLOP/s against 17 / 33 at N = 1024 / 1408, level at 1536, behind from 1664)
      if (tiles128 * 2 <= cus) return launch_valu_tile<64, 64, 64>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      return launch_valu_tile<128, 128, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_VALU_128X128:
      return launch_valu_tile<128, 128, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_VALU_64X64:
      return launch_valu_tile<64, 64, 64>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_NAIVE:
      return launch_naive(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_MFMA_SIMPLE:
      return launch_mfma<128, 128, true>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_MFMA_PIPE:
      return launch_mfma<128, 128, false, 0, 0, false>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_MFMA_256:
      return launch_mfma<256, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_AUTO: {
      // Tile choice by how well the shape fills 256 CUs (measured, profiles/r01_sweep.md):
      // with fewer 128x128 tiles than ~0.8 per CU the 128x64 configuration (twice the
      // workgroups, 94 % of the per-tile efficiency) wins; when even those number no more
      // than half the CUs, the 64x64 configuration; everything else is K2.  Each choice
      // runs as a stream-K launch when its tile count is ragged.
      const long tiles128x64 = (long)((m + 127) / 128) * ((n + 63) / 64);
      const long tiles256 = (long)((m + 255) / 256) * ((n + 255) / 256);
      // The 64x64 LDS-DMA tile with three workgroups co-resident per CU has the most efficient loop of
      // all (148.5-150 TFLOP/s at N = 3072, where 2304 tiles are exactly nine per CU; 151.6-152.9 at 5120 ..
      // 8192 against 148.5-150.4 for the 256x256 tile) -- on a PLAIN launch: under the chained stream-K
      // launch its workgroups run at different K phases and stop sharing operand slices in L2 (hit rate
      // 81 % -> 22 %, 2.4 GB of fabric traffic per launch, profiles/r02_ablation.md section 9).  So it is
      // chosen for whole-tile shapes with many tiles (>= 6 per CU) that fill their last round of CUs to
      // >= 97.5 % (N = 2688, 3072, 3200 on the reference sweep; 5120, 6144, 8192).
      // One exception: whole rounds of 256x256 tiles in a SHORT launch (N = 4096: one tile per CU, 0.93 ms).
      // Sustained the two are level there (148.5-149.4 vs 148.7-150.5), but from an idle clock the big
      // tile is within 1 % of its rate after 18 launches and the small one after 40 -- and 20 launches
      // from idle is what the reference's timing convention measures (137 vs 129 TFLOP/s,
      // profiles/r02_cold_start.txt).
      {
        const long tiles64 = (long)((m + 63) / 64) * ((n + 63) / 64);
        const long rounds64 = (tiles64 + cus - 1) / cus;
        const double est_ms = 2.0 * (double)m * (double)n * (double)k / 150e9;
        const bool whole_rounds_256 = tiles256 >= cus && tiles256 % cus == 0 && m % 256 == 0 && n % 256 == 0;
        // ... and only up to N = 8192-sized problems: with K = 16384 and B beyond the Infinity Cache the
        // small tile's two slices of look-ahead no longer cover its misses (2048 .. 16384 x 16384 x 16384:
        // 147.8 .. 140.0 against 150.9-151.1 for the 256x256 tile, which those shapes keep)
        if (!(whole_rounds_256 && est_ms < 2.0) && k <= 8192 && tiles64 <= 64 * cus && tiles64 >= 6 * cus &&
            tiles64 * 1000 >= rounds64 * cus * 975 &&
            window_ok(64, 64, k, lda, ldb) && fast_shape(64, 64, 32, m, n, k, dA, lda, dB, ldb, dC, ldc))
          return sgemm_on(ctx, MMH_KERNEL_MFMA_64X64_DMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
      }
      // At least one 256x256 tile per CU: the big tile (fewest staging ops per MFMA) -- unless its
      // edge tiles pad the shape noticeably more than 128x128 tiles would (an edge tile costs a whole
      // tile's time; ragged tile COUNTS are balanced by stream-K for either size).
      // a ragged count of 256x256 tiles would run as stream-K with ~1.1-1.2 tiles per workgroup; the 128x64
      // tile covers the same shape with >= 9 tiles per workgroup pair, phase-ordered (N = 4352 / 4608:
      // 148.6 / 148.9 against 147.2 / 147.4)
      if (tiles256 >= cus && tiles256 % cus != 0 && k <= 8192 && tiles128x64 <= 32 * cus &&
          tiles128x64 * 10 >= 2 * cus * 18 && window_ok(128, 128, k, lda, ldb) &&
          fast_shape(128, 64, 32, m, n, k, dA, lda, dB, ldb, dC, ldc))
        return sgemm_on(ctx, MMH_KERNEL_MFMA_128X64_DMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
      if (tiles256 >= cus) {
        const double fill256 = (double)m * (double)n / ((double)tiles256 * 65536.0);
        const double fill128 = (double)m * (double)n / ((double)tiles128 * 16384.0);
        if (fill256 >= fill128 - 0.015)
          return sgemm_on(ctx, MMH_KERNEL_MFMA_256X256, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
      }
      // OPT-IN split-K (default off: it gives up the one-chain-per-element bits, see sgemm_mfma.hpp K2s):
      // shapes with fewer 128x128 tiles than workgroup slots run their K ranges concurrently
      if (ctx && ctx->splitk > 0 && tiles128 < cus) {
        int S = ctx->splitk;
        if (S == 1) {   // auto: fill two workgroups per CU, keep >= 8 K-slices per part
          S = (int)((2 * cus) / (tiles128 > 0 ? tiles128 : 1));
          const int by_k = k / (8 * mmh::BK);
          if (S > by_k) S = by_k;
          if (S > 8) S = 8;
        }
        if (S >= 2) {
          const int sk = try_launch_splitk<128, 128, 4>(ctx, S, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
          if (sk <= 0) return sk;
        }
      }
      // Below one 256x256 tile per CU (N < 4096 on the reference sweep) whole-tile, 16-byte aligned
      // shapes run on the LDS-DMA tiles (sgemm_dma.hpp), the choice measured on the sweep
      // (profiles/r02_ablation.md): 128x128 from 1.15 tiles per CU (N >= 2304), 128x64 from 1.25 of
      // those per CU (N >= 1664), 64x64 below -- each as a chained stream-K launch when worthwhile.
      if (window_ok(128, 128, k, lda, ldb)) {
        // two co-resident 128x64 workgroups per CU beat one 128x128 workgroup by 1-1.5 % once the launch
        // is a phase-ordered stream-K (>= 1.8 tiles per workgroup of a 2-per-CU grid: N >= 2816)
        if (tiles128x64 * 10 >= 2 * cus * 18 && fast_shape(128, 64, 32, m, n, k, dA, lda, dB, ldb, dC, ldc))
          return sgemm_on(ctx, MMH_KERNEL_MFMA_128X64_DMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
        if (tiles128 * 100 >= cus * 115 && fast_shape(128, 128, 32, m, n, k, dA, lda, dB, ldb, dC, ldc))
          return sgemm_on(ctx, MMH_KERNEL_MFMA_128X128_DMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
        if (tiles128x64 * 100 >= cus * 125 && fast_shape(128, 64, 32, m, n, k, dA, lda, dB, ldb, dC, ldc))
          return sgemm_on(ctx, MMH_KERNEL_MFMA_128X64_DMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
        if (fast_shape(64, 64, 32, m, n, k, dA, lda, dB, ldb, dC, ldc))
          return sgemm_on(ctx, MMH_KERNEL_MFMA_64X64_DMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
      }
      if (tiles128x64 * 2 <= cus)
        return sgemm_on(ctx, MMH_KERNEL_MFMA_64X64, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
      if (tiles128 * 10 < cus * 8)
        return sgemm_on(ctx, MMH_KERNEL_MFMA_128X64, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
    }  // fall through
    case MMH_KERNEL_MFMA: {
      // ragged tile counts go to the persistent stream-K launch (same arithmetic,
      // same bits); everything else is one workgroup per tile
      const int sk = try_launch_streamk<128, 128, 4>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return launch_mfma<128, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    }
    case MMH_KERNEL_MFMA_TILES:   // K2 without stream-K (one workgroup per tile, always)
      return launch_mfma<128, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case MMH_KERNEL_MFMA_256X256: {  // 256x256 tile, 8 waves of 128x64 (one workgroup per CU)
      const int sk = try_launch_streamk<256, 256, 4, 8, 32>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return launch_mfma<256, 256, false, 4, 0, true, 4, 8, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    }
    case MMH_KERNEL_MFMA_64X64: {    // 64x64 tile, 4 waves of 32x32, 128-deep K-slices
      const int sk = try_launch_streamk<64, 64, 2, 2, 128>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return launch_mfma<64, 64, false, 4, 0, true, 2, 2, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    }
    case MMH_KERNEL_MFMA_128X64: {   // 128x64 tile, 4 waves of 64x32
      const int sk = try_launch_streamk<128, 64, 2>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return launch_mfma<128, 64, false, 4, 0, true, 2>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    }
    case MMH_KERNEL_MFMA_64X64_DMA: {   // K2L: 64x64 tile, both operands by LDS-DMA, 3 ring buffers of 32-deep slices (48 KiB: 3 WG/CU)
      const int sk = try_launch_streamk_dma<64, 64, 32, 2, 2, 3>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      const int d = try_launch_dma<64, 64, 32, 2, 2, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (d <= 0) return d;
      return sgemm_on(ctx, MMH_KERNEL_MFMA_64X64, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
    }
    case MMH_KERNEL_MFMA_128X64_DMA: {  // K2L on the 128x64 tile (4 waves of 64x32; 72 KiB ring: 2 WG/CU)
      const int sk = try_launch_streamk_dma<128, 64, 32, 4, 2, 3>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      const int d = try_launch_dma<128, 64, 32, 4, 2, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (d <= 0) return d;
      return sgemm_on(ctx, MMH_KERNEL_MFMA_128X64, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
    }
    case MMH_KERNEL_MFMA_128X128_DMA: {  // K2L on the 128x128 tile (4 waves of 64x64), 32-deep slices
      const int sk = try_launch_streamk_dma<128, 128, 32, 4, 4, 3>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      const int d = try_launch_dma<128, 128, 32, 4, 4, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (d <= 0) return d;
      return sgemm_on(ctx, MMH_KERNEL_MFMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
    }
    case MMH_KERNEL_MFMA_SPLITK: {   // K2s forced: 128x128 tiles, ctx->splitk parts (auto when <= 1)
      int S = ctx ? ctx->splitk : 0;
      if (S <= 1) {
        S = (int)((2 * cus) / (tiles128 > 0 ? tiles128 : 1));
        const int by_k = k / (8 * mmh::BK);
        S = std::min(std::min(S, by_k), 8);
      }
      const int sk = try_launch_splitk<128, 128, 4>(ctx, S, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return sgemm_on(ctx, MMH_KERNEL_MFMA, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
    }
    case MMH_KERNEL_MFMA_SPLITK_128X64: {   // K2s on 128x64 tiles
      int S = ctx ? ctx->splitk : 0;
      const long tiles = (long)((m + 127) / 128) * ((n + 63) / 64);
      if (S <= 1) {
        S = (int)((2 * cus) / (tiles > 0 ? tiles : 1));
        const int by_k = k / (8 * mmh::BK);
        S = std::min(std::min(S, by_k), 8);
      }
      const int sk = try_launch_splitk<128, 64, 2>(ctx, S, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return sgemm_on(ctx, MMH_KERNEL_MFMA_128X64, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
    }
#ifdef MMH_AB_BUILD
    // ---- tools-only variants (libmmult_hip_ab.so); never part of the product library ----
    case 19: {  // A/B: B through LDS-DMA (buffer_load ... lds)
      const int nbm = m / 128, nbn = n / 128;
      if (!fast_shape(128, 128, 32, m, n, k, dA, lda, dB, ldb, dC, ldc)) return MMH_ERR_INVALID_ARG;
      auto kern = mmh::sgemm_mfma_kernel<128, 128, false, 4, 0, true, 4, 4, 32, true>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * nbn)), dim3(256), lds_bytes(128, 128), s, m, n, k, dA,
                         lda, dB, ldb, dC, ldc, acc, nbm, nbn);
      HIP_TRY(hipGetLastError());
      return MMH_OK;
    }
    // ablation builds of the 256x256 configuration (TIMING ONLY): 21 no global loads, 22 + no LDS
    // stores, 23 + no barrier, 24 + no fragment reads (MFMAs only)
    case 21:
      return launch_mfma<256, 256, false, 4, 1, true, 4, 8, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 22:
      return launch_mfma<256, 256, false, 4, 3, true, 4, 8, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 23:
      return launch_mfma<256, 256, false, 4, 7, true, 4, 8, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 24:
      return launch_mfma<256, 256, false, 4, 15, true, 4, 8, 32>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 16:   // staging cadence A/B: one op per 3 / 4 MFMAs instead of 2
      return launch_mfma<128, 128, false, 5>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 17:
      return launch_mfma<128, 128, false, 6>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 18:
      return launch_mfma<128, 128, false, 7>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    // Ablation builds of the shipping kernel (TIMING ONLY -- results are wrong):
    // 32 no global loads, 33 + no LDS stores, 34 + no barrier, 35 + no fragment reads.
    case 32:
      return launch_mfma<128, 128, false, 4, 1>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 33:
      return launch_mfma<128, 128, false, 4, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 34:
      return launch_mfma<128, 128, false, 4, 7>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 35:
      return launch_mfma<128, 128, false, 4, 15>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    // the same for the 128x64 configuration (36 = loads always from the first two slices, i.e. cache-hot)
    case 36:
      return launch_mfma<128, 64, false, 4, 16, true, 2>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 37:
      return launch_mfma<128, 64, false, 4, 1, true, 2>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 38:
      return launch_mfma<128, 64, false, 4, 3, true, 2>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 39:
      return launch_mfma<128, 64, false, 4, 7, true, 2>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 40:
      return launch_mfma<128, 64, false, 4, 15, true, 2>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    // and for the 64x64 configuration (one wave per SIMD, 128-deep slices)
    case 41:
      return launch_mfma<64, 64, false, 4, 1, true, 2, 2, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 42:
      return launch_mfma<64, 64, false, 4, 3, true, 2, 2, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 43:
      return launch_mfma<64, 64, false, 4, 7, true, 2, 2, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    case 44:
      return launch_mfma<64, 64, false, 4, 15, true, 2, 2, 128>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
    // A/B (valid results): the LDS-DMA tiles as EIGHT waves -- two waves per SIMD from one workgroup
    case 45: {  // 64x64, waves of 16x32
      const int sk = try_launch_streamk_dma<64, 64, 32, 1, 2, 3>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return try_launch_dma<64, 64, 32, 1, 2, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    }
    case 46: {  // 128x64, waves of 32x32
      const int sk = try_launch_streamk_dma<128, 64, 32, 2, 2, 3>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return try_launch_dma<128, 64, 32, 2, 2, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    }
    case 47: {  // 128x128, waves of 64x32
      const int sk = try_launch_streamk_dma<128, 128, 32, 4, 2, 3>(ctx, m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
      if (sk <= 0) return sk;
      return try_launch_dma<128, 128, 32, 4, 2, 3>(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s) <= 0 ? MMH_OK : MMH_ERR_UNSUPPORTED;
    }
#endif
    default:
      g_last_error = "unknown kernel variant";
      return MMH_ERR_INVALID_ARG;
  }
}

bool is_gfx950(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int create_context(mmh_context **out, int device) {
  *out = nullptr;
  int count = 0;
  mmh_device_count(&count);
  if (count <= 0 || device < 0 || device >= count) {
    g_last_error = "no such HIP device";
    return MMH_ERR_NO_DEVICE;
  }
  if (!is_gfx950(device)) {
    g_last_error = "device is not gfx950 (this library carries gfx950 code objects only)";
    return MMH_ERR_NO_DEVICE;
  }
  DeviceGuard guard;
  HIP_TRY(guard.enter(device));
  mmh_context *ctx = new (std::nothrow) mmh_context;
  if (!ctx) return MMH_ERR_ALLOC;
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  if (const char *e = std::getenv("MMH_NO_PIN")) ctx->pin = (*e && *e != '0') ? 0 : 1;   // diagnostic A/B switches
  if (const char *e = std::getenv("MMH_NO_SK_ORDER")) ctx->sk_order = (*e && *e != '0') ? 0 : 1;
  // the sticky error word: pinned, mapped host memory (the device adds to it with a system-scope atomic)
  void *host = nullptr, *dev = nullptr;
  if (hipHostMalloc(&host, 64, hipHostMallocMapped) == hipSuccess) {
    memset(host, 0, 64);
    if (hipHostGetDevicePointer(&dev, host, 0) == hipSuccess) {
      ctx->sticky = static_cast<int *>(host);
      ctx->sticky_dev = static_cast<int *>(dev);
    } else {
      (void)hipHostFree(host);
    }
  }
  (void)hipGetLastError();   // without the word the persistent launches are simply not used
  *out = ctx;
  return MMH_OK;
}

void destroy_context(mmh_context *h) {
  if (!h) return;
  DeviceGuard guard;
  (void)guard.enter(h->device);
  h->a.release();
  h->b.release();
  h->c.release();
  h->flags.release();
  h->parts.release();
  h->bt.release();
  h->qa.release();
  h->qb.release();
  h->qc.release();
  h->qs.release();
  for (auto &t : h->sk_tables) t.buf.release();
  if (h->pipeline_ready) {
    for (int i = 0; i < kMaxHostPanels; ++i) {
      if (h->ev_in[i]) (void)hipEventDestroy(h->ev_in[i]);
      if (h->ev_run[i]) (void)hipEventDestroy(h->ev_run[i]);
    }
    if (h->ev_b) (void)hipEventDestroy(h->ev_b);
    if (h->hs_in) (void)hipStreamDestroy(h->hs_in);
    if (h->hs_run) (void)hipStreamDestroy(h->hs_run);
    if (h->hs_out) (void)hipStreamDestroy(h->hs_out);
  }
  if (h->t0) (void)hipEventDestroy(h->t0);
  if (h->t1) (void)hipEventDestroy(h->t1);
  if (h->sticky) (void)hipHostFree(h->sticky);
  mmh::rocblas_release(h->rocblas);
  delete h;
}

// ---- host flavour: row-panel pipeline --------------------------------------------------------
// The plain form moves A, B (and C when accumulating) in, runs the GEMM, moves C out, one after the
// other: at N = 4096 that is 5.8 ms of PCIe around a 0.93 ms kernel.  The pipelined form cuts A and C
// into row panels (mmh_shard_rows' 128-row granularity): after B, panel i's A (and C) go in on the
// copy-in stream, its GEMM runs on the compute stream as soon as they have landed, and its C rows go
// out on the copy-out stream -- from a helper thread, because a copy from/to pageable host memory
// blocks the calling thread -- while panel i+1 is still going in.  Row panels of C depend on nothing
// but their own rows of A (the same fact the multi-GPU shard rests on), so the bits are those of the
// single launch.  What is left is the H2D time of A, B (and C): PCIe is the floor of this flavour.
int ensure_pipeline(mmh_context *h) {
  if (h->pipeline_ready) return MMH_OK;
  HIP_TRY(hipStreamCreateWithFlags(&h->hs_in, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&h->hs_run, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&h->hs_out, hipStreamNonBlocking));
  for (int i = 0; i < kMaxHostPanels; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_run[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming));
  h->pipeline_ready = true;
  return MMH_OK;
}

int ensure_timing_events(mmh_context *h) {
  if (!h->t0) HIP_TRY(hipEventCreate(&h->t0));
  if (!h->t1) HIP_TRY(hipEventCreate(&h->t1));
  return MMH_OK;
}

int sgemm_host_pipelined(mmh_context *h, int panels, int m, int n, int k, const float *A, int lda, const float *B,
                         int ldb, float *C, int ldc, int accumulate, float *dA, float *dB, float *dC) {
  int rc = ensure_pipeline(h);
  if (rc != MMH_OK) return rc;
  struct Panel { int row0, rows; };
  std::vector<Panel> plan;
  for (int p = 0; p < panels; ++p) {
    Panel q{0, 0};
    mmh_shard_rows(m, panels, p, &q.row0, &q.rows);
    if (q.rows > 0) plan.push_back(q);
  }
  const int np = (int)plan.size();
  // copy-out helper: waits for panel i's GEMM, then moves its C rows to the host.  The events are
  // reused from call to call, so the helper first waits (mutex + condition variable) until THIS
  // call has recorded ev_run[i] -- an event still carrying last call's record would read "done".
  int out_rc = MMH_OK;           // written by the helper only, read after join()
  std::string out_err;
  std::atomic<bool> abandon{false};   // set by this thread when a later panel's GEMM will never run
  std::mutex mu;
  std::condition_variable cv;
  int recorded = 0;
  auto publish = [&](int upto) {
    { std::lock_guard<std::mutex> lock(mu); recorded = upto; }
    cv.notify_all();
  };
  std::thread out([&] {
    const bool dev_ok = hipSetDevice(h->device) == hipSuccess;
    if (!dev_ok) { out_rc = MMH_ERR_HIP; out_err = "hipSetDevice (copy-out thread)"; }
    for (int i = 0; i < np; ++i) {
      { std::unique_lock<std::mutex> lock(mu); cv.wait(lock, [&] { return recorded > i; }); }
      if (out_rc != MMH_OK || abandon.load()) continue;   // keep draining the hand-shake, copy nothing more
      hipError_t e = hipEventSynchronize(h->ev_run[i]);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(C + (size_t)plan[i].row0 * ldc, (size_t)ldc * 4, dC + (size_t)plan[i].row0 * n,
                             (size_t)n * 4, (size_t)n * 4, plan[i].rows, hipMemcpyDeviceToHost, h->hs_out);
      if (e == hipSuccess) e = hipStreamSynchronize(h->hs_out);
      if (e != hipSuccess) { out_rc = MMH_ERR_HIP; out_err = std::string("copy-out: ") + hipGetErrorString(e); }
    }
  });
  // Every ev_run[i] the helper waits for MUST be recorded, whatever fails in between: on an error the
  // remaining events are recorded on the (then idle) compute stream so that the helper drains.
  int issued = 0;
  auto finish = [&](int code) {
    if (code != MMH_OK) abandon.store(true);   // the helper must not copy panels whose GEMM never ran
    for (int i = issued; i < np; ++i) (void)hipEventRecord(h->ev_run[i], h->hs_run);
    publish(np);
    out.join();
    (void)hipStreamSynchronize(h->hs_in);
    (void)hipStreamSynchronize(h->hs_run);
    if (code == MMH_OK && out_rc != MMH_OK) {
      g_last_error = out_err;
      return out_rc;
    }
    return code;
  };
#define PIPE_TRY(expr)                                  \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return finish(hip_fail(e_, #expr)); \
  } while (0)
  PIPE_TRY(hipMemcpy2DAsync(dB, (size_t)n * 4, B, (size_t)ldb * 4, (size_t)n * 4, k, hipMemcpyHostToDevice, h->hs_in));
  PIPE_TRY(hipEventRecord(h->ev_b, h->hs_in));
  PIPE_TRY(hipStreamWaitEvent(h->hs_run, h->ev_b, 0));
  for (int i = 0; i < np; ++i) {
    const int r0 = plan[i].row0, rows = plan[i].rows;
    PIPE_TRY(hipMemcpy2DAsync(dA + (size_t)r0 * k, (size_t)k * 4, A + (size_t)r0 * lda, (size_t)lda * 4, (size_t)k * 4,
                              rows, hipMemcpyHostToDevice, h->hs_in));
    if (accumulate)
      PIPE_TRY(hipMemcpy2DAsync(dC + (size_t)r0 * n, (size_t)n * 4, C + (size_t)r0 * ldc, (size_t)ldc * 4,
                                (size_t)n * 4, rows, hipMemcpyHostToDevice, h->hs_in));
    PIPE_TRY(hipEventRecord(h->ev_in[i], h->hs_in));
    PIPE_TRY(hipStreamWaitEvent(h->hs_run, h->ev_in[i], 0));
    rc = sgemm_on(h, h->kernel, rows, n, k, dA + (size_t)r0 * k, k, dB, n, dC + (size_t)r0 * n, n, accumulate, h->hs_run);
    if (rc != MMH_OK) return finish(rc);
    PIPE_TRY(hipEventRecord(h->ev_run[i], h->hs_run));
    issued = i + 1;
    publish(issued);
  }
#undef PIPE_TRY
  return finish(MMH_OK);
}

}  // namespace

// ===========================================================================
struct mmh_shard {
  int ngpus = 0;
  int kernel = MMH_KERNEL_AUTO;
  int rccl_ranks = 0;                 // ranks of the communicator (0 when ngpus == 1: no RCCL)
  std::vector<int> devices;
  std::vector<mmh_context *> ctx;     // one product handle per device (stream-K workspaces etc.)
  std::vector<hipStream_t> streams;
  std::vector<DevBuf> a, b, c;        // per device: A panel, B, C panel
  std::vector<void *> comms;
};

extern "C" {

const char *mmh_strerror(int status) {
  switch (status) {
    case MMH_OK: return "success";
    case MMH_ERR_INVALID_ARG: return "invalid argument";
    case MMH_ERR_HIP: return "HIP runtime error";
    case MMH_ERR_NO_DEVICE: return "no gfx950 device";
    case MMH_ERR_UNSUPPORTED: return "unsupported in this build";
    case MMH_ERR_ALLOC: return "allocation failed";
    case MMH_ERR_COMM: return "RCCL error";
    default: return "unknown status";
  }
}

const char *mmh_last_error(void) { return g_last_error.c_str(); }

const char *mmh_last_launch(void) { return g_last_launch.c_str(); }

int mmh_version(void) { return 200; }

int mmh_is_ab_build(void) {
#ifdef MMH_AB_BUILD
  return 1;
#else
  return 0;
#endif
}

int mmh_device_count(int *count) {
  if (!count) return MMH_ERR_INVALID_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    (void)hipGetLastError();
    return MMH_OK;  // "no devices" is an answer, not a failure
  }
  *count = c;
  return MMH_OK;
}

int mmh_device_info(int device, char *name, int *cu_count, int *clock_mhz) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (name) snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
  return MMH_OK;
}

int mmh_create(mmh_handle_t *handle, int device) {
  if (!handle) return MMH_ERR_INVALID_ARG;
  return create_context(handle, device);
}

int mmh_destroy(mmh_handle_t h) {
  destroy_context(h);
  return MMH_OK;
}

int mmh_set_kernel(mmh_handle_t h, int kernel) {
  if (!h || !known_kernel(kernel)) return MMH_ERR_INVALID_ARG;
  h->kernel = kernel;
  return MMH_OK;
}

int mmh_set_option(mmh_handle_t h, int option, int value) {
  if (!h) return MMH_ERR_INVALID_ARG;
  switch (option) {
    case MMH_OPT_STREAMK:
      if (value < 0 || value > 2) return MMH_ERR_INVALID_ARG;
      h->streamk = value;
      return MMH_OK;
    case MMH_OPT_STREAMK_TIMEOUTS:   // writing 0 clears the sticky error
      if (value != 0) return MMH_ERR_INVALID_ARG;
      {
        DeviceGuard guard;
        HIP_TRY(guard.enter(h->device));
        HIP_TRY(hipDeviceSynchronize());
      }
      if (h->sticky) *reinterpret_cast<volatile int *>(h->sticky) = 0;
      h->flags_dirty = true;   // a launch that timed out may have left hand-off counters behind
      return MMH_OK;
    case MMH_OPT_IGEMM_MODE:
      if ((value >= 0 && value <= 6)
#ifdef MMH_AB_BUILD
          || (value >= 10 && value <= 13)
#endif
      ) {
        h->igemm_mode = value;
        return MMH_OK;
      }
      return MMH_ERR_INVALID_ARG;
    case MMH_OPT_SPLITK:
      if (value < 0 || value > 16) return MMH_ERR_INVALID_ARG;
      h->splitk = value;
      return MMH_OK;
    case MMH_OPT_HOST_PANELS:
      if (value < -1 || value > kMaxHostPanels) return MMH_ERR_INVALID_ARG;
      h->host_panels = value;
      return MMH_OK;
    case MMH_OPT_STREAMK_SPIN_LIMIT:   // in units of 1024 polls
      if (value < 1) return MMH_ERR_INVALID_ARG;
      h->spin_limit = (long long)value << 10;
      return MMH_OK;
    case MMH_OPT_FAULT_INJECT:
      h->fault = value ? 1 : 0;
      h->flags_dirty = true;
      return MMH_OK;
    case MMH_OPT_STREAMK_ORDER:
      h->sk_order = value ? 1 : 0;
      return MMH_OK;
#ifdef MMH_AB_BUILD
    case 100:   // A/B: pin the residency of persistent launches by their LDS request (default on)
      h->pin = value ? 1 : 0;
      return MMH_OK;
#endif
    default:
      return MMH_ERR_INVALID_ARG;
  }
}

int mmh_get_option(mmh_handle_t h, int option, int *value) {
  if (!h || !value) return MMH_ERR_INVALID_ARG;
  switch (option) {
    case MMH_OPT_STREAMK: *value = h->streamk; return MMH_OK;
    case MMH_OPT_IGEMM_MODE: *value = h->igemm_mode; return MMH_OK;
    case MMH_OPT_SPLITK: *value = h->splitk; return MMH_OK;
    case MMH_OPT_HOST_PANELS: *value = h->host_panels; return MMH_OK;
    case MMH_OPT_STREAMK_SPIN_LIMIT: *value = (int)(h->spin_limit >> 10); return MMH_OK;
    case MMH_OPT_FAULT_INJECT: *value = h->fault; return MMH_OK;
    case MMH_OPT_STREAMK_ORDER: *value = h->sk_order; return MMH_OK;
    case MMH_OPT_STREAMK_TIMEOUTS: {
      // synchronises, then reads the sticky word: how many hand-off waits have timed out on this
      // handle since it was last cleared
      *value = 0;
      DeviceGuard guard;
      HIP_TRY(guard.enter(h->device));
      HIP_TRY(hipDeviceSynchronize());
      if (h->sticky) *value = *reinterpret_cast<volatile int *>(h->sticky);
      return MMH_OK;
    }
    default:
      return MMH_ERR_INVALID_ARG;
  }
}

int mmh_get_kernel(mmh_handle_t h, int *kernel) {
  if (!h || !kernel) return MMH_ERR_INVALID_ARG;
  *kernel = h->kernel;
  return MMH_OK;
}

const char *mmh_kernel_name(int kernel) {
  switch (kernel) {
    case MMH_KERNEL_AUTO: return "MMult_hip_auto";
    case MMH_KERNEL_VALU: return "MMult_hip_valu";
    case MMH_KERNEL_VALU_128X128: return "MMult_hip_valu_128x128";
    case MMH_KERNEL_VALU_64X64: return "MMult_hip_valu_64x64";
    case MMH_KERNEL_MFMA: return "MMult_hip_mfma";
    case MMH_KERNEL_MFMA_256: return "MMult_hip_mfma256";
    case MMH_KERNEL_NAIVE: return "MMult_hip_naive";
    case MMH_KERNEL_MFMA_SIMPLE: return "MMult_hip_mfma_simple";
    case MMH_KERNEL_MFMA_PIPE: return "MMult_hip_mfma_pipe";
    case MMH_KERNEL_MFMA_TILES: return "MMult_hip_mfma_tiles";
    case MMH_KERNEL_MFMA_128X64: return "MMult_hip_mfma_128x64";
    case MMH_KERNEL_MFMA_64X64: return "MMult_hip_mfma_64x64";
    case MMH_KERNEL_MFMA_256X256: return "MMult_hip_mfma_256x256";
    case MMH_KERNEL_MFMA_64X64_DMA: return "MMult_hip_mfma_64x64_dma";
    case MMH_KERNEL_MFMA_128X64_DMA: return "MMult_hip_mfma_128x64_dma";
    case MMH_KERNEL_MFMA_128X128_DMA: return "MMult_hip_mfma_128x128_dma";
    case MMH_KERNEL_MFMA_SPLITK: return "MMult_hip_mfma_splitk";
    case MMH_KERNEL_MFMA_SPLITK_128X64: return "MMult_hip_mfma_splitk_128x64";
#ifdef MMH_AB_BUILD
    case 19: return "exp_dma_b";
    case 16: return "cadence_3";
    case 17: return "cadence_4";
    case 18: return "cadence_1";
    case 32: return "ablate_no_gload";
    case 33: return "ablate_no_gload_no_ldswrite";
    case 34: return "ablate_no_gload_no_ldswrite_no_barrier";
    case 35: return "ablate_mfma_only";
    case 21: return "ablate256_no_gload";
    case 22: return "ablate256_no_gload_no_ldswrite";
    case 23: return "ablate256_no_gload_no_ldswrite_no_barrier";
    case 24: return "ablate256_mfma_only";
    case 36: return "ablate128x64_hot_loads";
    case 37: return "ablate128x64_no_gload";
    case 38: return "ablate128x64_no_gload_no_ldswrite";
    case 39: return "ablate128x64_no_gload_no_ldswrite_no_barrier";
    case 40: return "ablate128x64_mfma_only";
    case 41: return "ablate64x64_no_gload";
    case 42: return "ablate64x64_no_gload_no_ldswrite";
    case 43: return "ablate64x64_no_gload_no_ldswrite_no_barrier";
    case 44: return "ablate64x64_mfma_only";
    case 45: return "exp_dma_64x64_8waves";
    case 46: return "exp_dma_128x64_8waves";
    case 47: return "exp_dma_128x128_8waves";
#endif
    default: return nullptr;
  }
}

int mmh_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
              int ldb, float *dC, int ldc, int accumulate, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate,
                  static_cast<hipStream_t>(stream));
}

int mmh_sgemm_host(mmh_handle_t h, int m, int n, int k, const float *A, int lda, const float *B,
                   int ldb, float *C, int ldc, int accumulate) {
  return mmh_sgemm_host_timed(h, m, n, k, A, lda, B, ldb, C, ldc, accumulate, nullptr);
}

int mmh_sgemm_host_timed(mmh_handle_t h, int m, int n, int k, const float *A, int lda, const float *B,
                         int ldb, float *C, int ldc, int accumulate, float *kernel_ms) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, A, lda, B, ldb, C, ldc);
  if (rc != MMH_OK) return rc;
  if (kernel_ms) *kernel_ms = 0.0f;
  if (m == 0 || n == 0) return MMH_OK;
  ENTER(h);
  // Device images are dense (lda=k, ldb=n, ldc=n) whatever the host strides.
  const size_t ab = (size_t)m * k * sizeof(float), bb = (size_t)k * n * sizeof(float),
               cb = (size_t)m * n * sizeof(float);
  if ((rc = h->a.reserve(ab ? ab : 16)) != MMH_OK) return rc;
  if ((rc = h->b.reserve(bb ? bb : 16)) != MMH_OK) return rc;
  if ((rc = h->c.reserve(cb)) != MMH_OK) return rc;
  float *dA = static_cast<float *>(h->a.p), *dB = static_cast<float *>(h->b.p),
        *dC = static_cast<float *>(h->c.p);
  // row-panel pipeline when the problem is large enough for the copies to matter (>= 2 panels of
  // >= 512 rows and >= 16 MiB moved), unless MMH_OPT_HOST_PANELS says otherwise
  int panels = h->host_panels;
  if (panels < 0) {
    panels = 0;
    if (k > 0 && m >= 1024 && (ab + bb + cb) >= (16u << 20)) panels = std::min(8, m / 512);
  }
  // the timed form wants the device time of the GEMM alone (what the Vulkan flavour's timestamps
  // bracket, vulkan/MMult_vk_3.cpp:38-46): one launch between two events, no overlapping copies
  if (!kernel_ms && panels >= 2 && k > 0 && m >= 2 * 128)
    return sgemm_host_pipelined(h, std::min(panels, kMaxHostPanels), m, n, k, A, lda, B, ldb, C, ldc, accumulate, dA,
                                dB, dC);
  if (k > 0) {
    HIP_TRY(hipMemcpy2D(dA, (size_t)k * 4, A, (size_t)lda * 4, (size_t)k * 4, m,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy2D(dB, (size_t)n * 4, B, (size_t)ldb * 4, (size_t)n * 4, k,
                        hipMemcpyHostToDevice));
  }
  if (accumulate)
    HIP_TRY(hipMemcpy2D(dC, (size_t)n * 4, C, (size_t)ldc * 4, (size_t)n * 4, m,
                        hipMemcpyHostToDevice));
  if (kernel_ms) {
    if ((rc = ensure_timing_events(h)) != MMH_OK) return rc;
    HIP_TRY(hipEventRecord(h->t0, nullptr));
  }
  rc = sgemm_on(h, h->kernel, m, n, k, dA, k, dB, n, dC, n, accumulate, nullptr);
  if (rc != MMH_OK) return rc;
  if (kernel_ms) HIP_TRY(hipEventRecord(h->t1, nullptr));
  HIP_TRY(hipMemcpy2D(C, (size_t)ldc * 4, dC, (size_t)n * 4, (size_t)n * 4, m,
                      hipMemcpyDeviceToHost));
  if (kernel_ms) {
    HIP_TRY(hipEventSynchronize(h->t1));
    HIP_TRY(hipEventElapsedTime(kernel_ms, h->t0, h->t1));
  }
  return MMH_OK;
}

int mmh_igemm_s8(mmh_handle_t h, int m, int n, int k, const int8_t *dA, int lda, const int8_t *dB,
                 int ldb, int32_t *dC, int ldc, int accumulate, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k == 0) {
    if (!accumulate)
      HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * 4, 0, (size_t)n * 4, (size_t)m, s));
    return MMH_OK;
  }
  // Default mode: operands the in-place kernel cannot take as they are (an odd leading dimension, a
  // base that is not dword-aligned) are first copied into dense dword-aligned workspace images -- one
  // pass over m*k / k*n bytes, against m*n*k MACs -- instead of falling back to the slow kernels.
  if (h->igemm_mode == 0 && !mmh::igemm_s8_inplace_ok(dA, lda, dB, ldb, k)) {
    const bool a_ok = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(dA) & 3) == 0);
    const bool b_ok = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(dB) & 3) == 0);
    const int ka = a_ok ? lda : (k + 15) & ~15, nb = b_ok ? ldb : (n + 15) & ~15;
    const int8_t *sa = dA, *sb = dB;
    if (!a_ok) {
      if ((rc = h->qa.reserve((size_t)m * ka)) != MMH_OK) return rc;
      HIP_TRY(hipMemcpy2DAsync(h->qa.p, (size_t)ka, dA, (size_t)lda, (size_t)k, (size_t)m, hipMemcpyDeviceToDevice, s));
      sa = static_cast<const int8_t *>(h->qa.p);
    }
    if (!b_ok) {
      if ((rc = h->qb.reserve((size_t)k * nb)) != MMH_OK) return rc;
      HIP_TRY(hipMemcpy2DAsync(h->qb.p, (size_t)nb, dB, (size_t)ldb, (size_t)n, (size_t)k, hipMemcpyDeviceToDevice, s));
      sb = static_cast<const int8_t *>(h->qb.p);
    }
    if (mmh::igemm_s8_inplace_ok(sa, ka, sb, nb, k)) {
      HIP_TRY(mmh::launch_igemm_s8(m, n, k, sa, ka, sb, nb, dC, ldc, accumulate ? 1 : 0, s, nullptr, 0,
                                   h->cu_count > 0 ? h->cu_count : 256));
      return MMH_OK;
    }
    // (operands beyond the descriptors' 2 GiB window: the general path below)
  }
  int8_t *bt = nullptr;
  if (mmh::igemm_s8_needs_pack(h->igemm_mode, dA, lda, dB, ldb, k) &&
      h->bt.reserve(mmh::igemm_s8_pack_bytes(n, k)) == MMH_OK)
    bt = static_cast<int8_t *>(h->bt.p);
  HIP_TRY(mmh::launch_igemm_s8(m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate ? 1 : 0, s, bt, h->igemm_mode,
                               h->cu_count > 0 ? h->cu_count : 256));
  return MMH_OK;
}

int mmh_quantize_sym_s8(mmh_handle_t h, int rows, int cols, const float *dX, int ldx, int8_t *dQ,
                        int ldq, float *d_scale, void *stream) {
  if (!h || rows < 0 || cols < 0) return MMH_ERR_INVALID_ARG;
  if (rows == 0 || cols == 0) return MMH_OK;
  if (!dX || !dQ || !d_scale || ldx < cols || ldq < cols) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // qs: [0, 2 AMAX_WORDS) abs-max words of a quantised GEMM's A and B, then its two scales, then the
  // abs-max words of stand-alone calls
  constexpr size_t qs_words = 4 * mmh::AMAX_WORDS + 16;
  int rc = h->qs.reserve(qs_words * sizeof(unsigned));
  if (rc != MMH_OK) return rc;
  unsigned *amax = static_cast<unsigned *>(h->qs.p) + 2 * mmh::AMAX_WORDS + 16;
  HIP_TRY(hipMemsetAsync(amax, 0, 2 * mmh::AMAX_WORDS * sizeof(unsigned), s));
  const mmh::QuantTensor t{dX, rows, cols, ldx, dQ, ldq}, none{nullptr, 0, 0, 0, nullptr, 0};
  const bool v_in = mmh::quant_vec_ok(t, false), v_out = mmh::quant_vec_ok(t, true);
  hipLaunchKernelGGL(mmh::absmax_kernel, dim3(mmh::quant_grid(t, v_in), 1), dim3(256), 0, s, t, none, v_in ? 1 : 0, 0, amax);
  hipLaunchKernelGGL(mmh::quantize_kernel, dim3(mmh::quant_grid(t, v_out), 1), dim3(256), 0, s, t, none, v_out ? 1 : 0, 0,
                     amax, d_scale);
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

int mmh_qgemm_f32(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
                  int ldb, float *dC, int ldc, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k == 0) {
    HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * 4, 0, (size_t)n * 4, (size_t)m, s));
    return MMH_OK;
  }
  // dense, 16-byte-friendly workspace images: int8 A (m x ka), int8 B (k x nb), int32 C (m x nb)
  const int ka = (k + 15) & ~15, nb = (n + 3) & ~3;
  if ((rc = h->qa.reserve((size_t)m * ka)) != MMH_OK) return rc;
  if ((rc = h->qb.reserve((size_t)k * nb)) != MMH_OK) return rc;
  constexpr size_t qs_words = 4 * mmh::AMAX_WORDS + 16;
  if ((rc = h->qs.reserve(qs_words * sizeof(unsigned))) != MMH_OK) return rc;
  int8_t *qa = static_cast<int8_t *>(h->qa.p), *qb = static_cast<int8_t *>(h->qb.p);
  unsigned *amax = static_cast<unsigned *>(h->qs.p);                           // A's words, then B's
  float *scales = reinterpret_cast<float *>(amax + 2 * mmh::AMAX_WORDS);       // [0] A, [1] B
  HIP_TRY(hipMemsetAsync(amax, 0, 2 * mmh::AMAX_WORDS * sizeof(unsigned), s));
  // A and B share one abs-max launch and one quantisation launch (blockIdx.y picks the tensor)
  const mmh::QuantTensor ta{dA, m, k, lda, qa, ka}, tb{dB, k, n, ldb, qb, nb};
  const bool va_in = mmh::quant_vec_ok(ta, false), vb_in = mmh::quant_vec_ok(tb, false);
  const bool va_out = mmh::quant_vec_ok(ta, true), vb_out = mmh::quant_vec_ok(tb, true);
  const dim3 gmax(std::max(mmh::quant_grid(ta, va_in), mmh::quant_grid(tb, vb_in)), 2);
  const dim3 g(std::max(mmh::quant_grid(ta, va_out), mmh::quant_grid(tb, vb_out)), 2);
  hipLaunchKernelGGL(mmh::absmax_kernel, gmax, dim3(256), 0, s, ta, tb, va_in ? 1 : 0, vb_in ? 1 : 0, amax);
  hipLaunchKernelGGL(mmh::quantize_kernel, g, dim3(256), 0, s, ta, tb, va_out ? 1 : 0, vb_out ? 1 : 0, amax, scales);
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  if (h->igemm_mode == 0 && mmh::igemm_s8_inplace_ok(qa, ka, qb, nb, k)) {
    // the int8 GEMM dequantises in its epilogue: no int32 image of C at all
    HIP_TRY(mmh::launch_igemm_s8_dequant(m, n, k, qa, ka, qb, nb, dC, ldc, scales, s, cus));
    return MMH_OK;
  }
  // two-pass form (A/B modes of the int8 kernel): int32 C, then the dequantisation pass
  if ((rc = h->qc.reserve((size_t)m * nb * sizeof(int32_t))) != MMH_OK) return rc;
  int32_t *qc = static_cast<int32_t *>(h->qc.p);
  int8_t *bt = nullptr;
  if (mmh::igemm_s8_needs_pack(h->igemm_mode, qa, ka, qb, nb, k) &&
      h->bt.reserve(mmh::igemm_s8_pack_bytes(n, k)) == MMH_OK)
    bt = static_cast<int8_t *>(h->bt.p);
  HIP_TRY(mmh::launch_igemm_s8(m, n, k, qa, ka, qb, nb, qc, nb, 0, s, bt, h->igemm_mode, cus));
  hipLaunchKernelGGL(mmh::dequantize_kernel, dim3(mmh::quant_rows_grid(m, 0)), dim3(256), 0, s, qc, m, n, nb,
                     scales, scales + 1, dC, ldc);
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

int mmh_sgemm_rocblas(mmh_handle_t h, int m, int n, int k, const float *dA, int lda,
                      const float *dB, int ldb, float *dC, int ldc, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  ENTER(h);
  if (m == 0 || n == 0 || k == 0) return sgemm_on(h, MMH_KERNEL_MFMA, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, static_cast<hipStream_t>(stream));
  return mmh::rocblas_sgemm_rowmajor(&h->rocblas, m, n, k, dA, lda, dB, ldb, dC, ldc, stream,
                                     &g_last_error);
}

int mmh_shard_rows(int m, int nranks, int rank, int *row0, int *rows) {
  if (m < 0 || nranks <= 0 || rank < 0 || rank >= nranks || !row0 || !rows)
    return MMH_ERR_INVALID_ARG;
  // Whole 128-row tiles first, dealt as evenly as possible from rank 0 up;
  // the ragged tail (m % 128 rows) rides with the last rank that has tiles
  // (or rank 0 if there are none), so every boundary is tile-aligned.
  const int tiles = m / 128, tail = m % 128;
  const int base = tiles / nranks, extra = tiles % nranks;
  const int my_tiles = base + (rank < extra ? 1 : 0);
  const int first_tile = rank * base + (rank < extra ? rank : extra);
  int r0 = first_tile * 128, nr = my_tiles * 128;
  int last_with_tiles = tiles == 0 ? 0 : (tiles >= nranks ? nranks - 1 : tiles - 1);
  if (rank == last_with_tiles) nr += tail;
  if (rank > last_with_tiles) r0 = m;  // empty panels sit at the end
  *row0 = r0;
  *rows = nr;
  return MMH_OK;
}

// ---- single-process row-panel shard: a handle (BASELINE.json config 4) ----------------------------
int mmh_rccl_version(int *version) {
  if (!version) return MMH_ERR_INVALID_ARG;
  *version = 0;
  mmh::RcclApi &api = mmh::rccl_api();
  if (!api.ok) {
    g_last_error = "librccl.so could not be loaded (or lacks an entry point the shard needs)";
    return MMH_ERR_UNSUPPORTED;
  }
  if (api.get_version(version) != 0) return MMH_ERR_COMM;
  return MMH_OK;
}

int mmh_shard_destroy(mmh_shard_t sh) {
  if (!sh) return MMH_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (int d = 0; d < (int)sh->devices.size(); ++d) {
    (void)hipSetDevice(sh->devices[d]);
    if (d < (int)sh->comms.size() && sh->comms[d]) mmh::rccl_api().comm_destroy(sh->comms[d]);
    if (d < (int)sh->a.size()) { sh->a[d].release(); sh->b[d].release(); sh->c[d].release(); }
    if (d < (int)sh->streams.size() && sh->streams[d]) (void)hipStreamDestroy(sh->streams[d]);
    if (d < (int)sh->ctx.size()) destroy_context(sh->ctx[d]);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  delete sh;
  return MMH_OK;
}

int mmh_shard_create(mmh_shard_t *out, int ngpus, const int *devices) {
  if (!out) return MMH_ERR_INVALID_ARG;
  *out = nullptr;
  if (ngpus <= 0 || ngpus > 64) return MMH_ERR_INVALID_ARG;
  int count = 0;
  mmh_device_count(&count);
  if (count < ngpus) {
    g_last_error = "fewer visible devices (" + std::to_string(count) + ") than ngpus (" + std::to_string(ngpus) + ")";
    return MMH_ERR_NO_DEVICE;
  }
  if (ngpus > 1 && !mmh::rccl_api().ok) {
    g_last_error = "librccl.so could not be loaded";
    return MMH_ERR_UNSUPPORTED;
  }
  mmh_shard *sh = new (std::nothrow) mmh_shard;
  if (!sh) return MMH_ERR_ALLOC;
  sh->ngpus = ngpus;
  for (int d = 0; d < ngpus; ++d) {
    const int dev = devices ? devices[d] : d;
    if (dev < 0 || dev >= count || std::find(sh->devices.begin(), sh->devices.end(), dev) != sh->devices.end()) {
      delete sh;
      g_last_error = "device list names a device twice or out of range";
      return MMH_ERR_INVALID_ARG;
    }
    sh->devices.push_back(dev);
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  sh->ctx.assign(ngpus, nullptr);
  sh->streams.assign(ngpus, nullptr);
  sh->a.resize(ngpus);
  sh->b.resize(ngpus);
  sh->c.resize(ngpus);
  sh->comms.assign(ngpus, nullptr);
  int rc = MMH_OK;
  for (int d = 0; d < ngpus && rc == MMH_OK; ++d) {
    rc = create_context(&sh->ctx[d], sh->devices[d]);
    if (rc != MMH_OK) break;
    if (hipSetDevice(sh->devices[d]) != hipSuccess || hipStreamCreate(&sh->streams[d]) != hipSuccess) {
      g_last_error = "hipStreamCreate failed";
      rc = MMH_ERR_HIP;
    }
  }
  if (rc == MMH_OK && ngpus > 1) {
    // ONE communicator for the life of the handle (creating it costs far more than any GEMM here)
    if (mmh::rccl_api().comm_init_all(sh->comms.data(), ngpus, sh->devices.data()) != 0) {
      g_last_error = "ncclCommInitAll failed";
      rc = MMH_ERR_COMM;
    } else {
      int ranks = 0;
      if (mmh::rccl_api().comm_count(sh->comms[0], &ranks) == 0) sh->rccl_ranks = ranks;
    }
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc != MMH_OK) {
    mmh_shard_destroy(sh);
    return rc;
  }
  *out = sh;
  return MMH_OK;
}

int mmh_shard_set_kernel(mmh_shard_t sh, int kernel) {
  if (!sh || !known_kernel(kernel)) return MMH_ERR_INVALID_ARG;
  sh->kernel = kernel;
  return MMH_OK;
}

int mmh_shard_info(mmh_shard_t sh, int *ngpus, int *rccl_ranks) {
  if (!sh) return MMH_ERR_INVALID_ARG;
  if (ngpus) *ngpus = sh->ngpus;
  if (rccl_ranks) *rccl_ranks = sh->rccl_ranks;
  return MMH_OK;
}

int mmh_shard_sgemm(mmh_shard_t sh, int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C,
                    int ldc, int gemm_reps, float *timings_ms) {
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<float, std::milli>(clk::now() - t).count(); };
  if (!sh || gemm_reps < 1) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, A, lda, B, ldb, C, ldc);
  if (rc != MMH_OK) return rc;
  if (timings_ms) timings_ms[0] = timings_ms[1] = timings_ms[2] = timings_ms[3] = 0.f;
  if (m == 0 || n == 0) return MMH_OK;
  const int G = sh->ngpus;
  for (int d = 0; d < G; ++d)
    if ((rc = check_sticky(sh->ctx[d])) != MMH_OK) return rc;
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore {
    int prev;
    ~Restore() { if (prev >= 0) (void)hipSetDevice(prev); }
  } restore{prev};
  std::vector<int> row0(G), rows(G);
  const size_t kk = k > 0 ? k : 1;
  for (int d = 0; d < G; ++d) {
    mmh_shard_rows(m, G, d, &row0[d], &rows[d]);
    HIP_TRY(hipSetDevice(sh->devices[d]));
    const size_t r = rows[d] > 0 ? rows[d] : 1;
    if ((rc = sh->a[d].reserve(r * kk * sizeof(float))) != MMH_OK) return rc;
    if ((rc = sh->b[d].reserve(kk * n * sizeof(float))) != MMH_OK) return rc;
    if ((rc = sh->c[d].reserve(r * n * sizeof(float))) != MMH_OK) return rc;
  }
  // One host thread per device for the host <-> device phases: a copy from/to pageable memory blocks
  // its calling thread, and every device has a PCIe link of its own.
  std::vector<hipError_t> err(G, hipSuccess);
  auto per_device = [&](auto &&fn) {
    std::vector<std::thread> pool;
    for (int d = 0; d < G; ++d)
      pool.emplace_back([&, d] {
        hipError_t e = hipSetDevice(sh->devices[d]);
        if (e == hipSuccess) e = fn(d);
        if (e == hipSuccess) e = hipStreamSynchronize(sh->streams[d]);
        err[d] = e;
      });
    for (auto &t : pool) t.join();
    for (int d = 0; d < G; ++d)
      if (err[d] != hipSuccess) return hip_fail(err[d], "row-panel shard: host <-> device phase");
    return (int)MMH_OK;
  };
  // ---- host -> device: A panels to their owners, B to device 0 only (every device when there is no
  // communicator, i.e. G == 1) ----
  auto t = clk::now();
  if (k > 0) {
    rc = per_device([&](int d) -> hipError_t {
      hipError_t e = hipSuccess;
      if (rows[d] > 0)
        e = hipMemcpy2DAsync(sh->a[d].p, (size_t)k * 4, A + (size_t)row0[d] * lda, (size_t)lda * 4, (size_t)k * 4,
                             rows[d], hipMemcpyHostToDevice, sh->streams[d]);
      if (e == hipSuccess && d == 0)
        e = hipMemcpy2DAsync(sh->b[0].p, (size_t)n * 4, B, (size_t)ldb * 4, (size_t)n * 4, k, hipMemcpyHostToDevice,
                             sh->streams[0]);
      return e;
    });
    if (rc != MMH_OK) return rc;
  }
  if (timings_ms) timings_ms[0] = ms_since(t);
  // ---- the one collective: broadcast B from device 0 over xGMI ----
  t = clk::now();
  if (G > 1 && k > 0) {
    mmh::RcclApi &api = mmh::rccl_api();
    bool bad = api.group_start() != 0;
    for (int d = 0; d < G && !bad; ++d) {
      constexpr int nccl_float = 7;
      bad = api.broadcast(sh->b[0].p, sh->b[d].p, (size_t)k * n, nccl_float, 0, sh->comms[d], sh->streams[d]) != 0;
    }
    if (api.group_end() != 0) bad = true;
    if (bad) {
      g_last_error = "ncclBroadcast failed";
      return MMH_ERR_COMM;
    }
    for (int d = 0; d < G; ++d) {
      HIP_TRY(hipSetDevice(sh->devices[d]));
      HIP_TRY(hipStreamSynchronize(sh->streams[d]));
    }
  }
  if (timings_ms) timings_ms[1] = (G > 1 && k > 0) ? ms_since(t) : 0.f;
  // ---- independent row-panel GEMMs (gemm_reps back-to-back launches per device: phase time / reps) ----
  t = clk::now();
  for (int rep = 0; rep < gemm_reps; ++rep)
    for (int d = 0; d < G; ++d) {
      if (rows[d] == 0) continue;
      HIP_TRY(hipSetDevice(sh->devices[d]));
      rc = sgemm_on(sh->ctx[d], sh->kernel, rows[d], n, k, static_cast<float *>(sh->a[d].p), k,
                    static_cast<float *>(sh->b[d].p), n, static_cast<float *>(sh->c[d].p), n, 0, sh->streams[d]);
      if (rc != MMH_OK) return rc;
    }
  for (int d = 0; d < G; ++d) {
    HIP_TRY(hipSetDevice(sh->devices[d]));
    HIP_TRY(hipStreamSynchronize(sh->streams[d]));
  }
  if (timings_ms) timings_ms[2] = ms_since(t) / gemm_reps;
  for (int d = 0; d < G; ++d)
    if ((rc = check_sticky(sh->ctx[d])) != MMH_OK) return rc;
  // ---- device -> host: disjoint C panels ----
  t = clk::now();
  rc = per_device([&](int d) -> hipError_t {
    if (rows[d] == 0) return hipSuccess;
    return hipMemcpy2DAsync(C + (size_t)row0[d] * ldc, (size_t)ldc * 4, sh->c[d].p, (size_t)n * 4, (size_t)n * 4,
                            rows[d], hipMemcpyDeviceToHost, sh->streams[d]);
  });
  if (rc != MMH_OK) return rc;
  if (timings_ms) timings_ms[3] = ms_since(t);
  return MMH_OK;
}

// one-shot convenience form: create, run once, destroy (what the round-1 entry point did on every call)
int mmh_sgemm_sharded(int ngpus, int m, int n, int k, const float *A, int lda, const float *B,
                      int ldb, float *C, int ldc, int kernel, float *timings_ms) {
  int rc = check_gemm_args(m, n, k, A, lda, B, ldb, C, ldc);
  if (rc != MMH_OK) return rc;
  if (ngpus <= 0 || !known_kernel(kernel)) return MMH_ERR_INVALID_ARG;
  mmh_shard_t sh = nullptr;
  if ((rc = mmh_shard_create(&sh, ngpus, nullptr)) != MMH_OK) return rc;
  rc = mmh_shard_set_kernel(sh, kernel);
  if (rc == MMH_OK) rc = mmh_shard_sgemm(sh, m, n, k, A, lda, B, ldb, C, ldc, 1, timings_ms);
  mmh_shard_destroy(sh);
  return rc;
}

int mmh_time_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
                   int ldb, float *dC, int ldc, int warmup, int reps, void *stream,
                   float *ms_per_call) {
  if (!h || reps <= 0 || warmup < 0 || !ms_per_call) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  for (int i = 0; i < warmup; ++i)
    if ((rc = sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, s)) != MMH_OK) return rc;
  hipEvent_t t0, t1;
  HIP_TRY(hipEventCreate(&t0));
  HIP_TRY(hipEventCreate(&t1));
  HIP_TRY(hipEventRecord(t0, s));
  for (int i = 0; i < reps; ++i)
    if ((rc = sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, s)) != MMH_OK) return rc;
  HIP_TRY(hipEventRecord(t1, s));
  HIP_TRY(hipEventSynchronize(t1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  *ms_per_call = ms / reps;
  return check_sticky(h);
}

// per-launch durations of `count` back-to-back calls (one event pair each): the clock-ramp trace
int mmh_trace_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                    float *dC, int ldc, int count, void *stream, float *ms_each) {
  if (!h || count <= 0 || count > 4096 || !ms_each) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(count + 1);
  for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
  int rc = MMH_OK;
  HIP_TRY(hipEventRecord(ev[0], s));
  for (int i = 0; i < count && rc == MMH_OK; ++i) {
    rc = sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, s);
    if (rc == MMH_OK && hipEventRecord(ev[i + 1], s) != hipSuccess) rc = MMH_ERR_HIP;
  }
  if (rc == MMH_OK && hipEventSynchronize(ev[count]) != hipSuccess) rc = MMH_ERR_HIP;
  for (int i = 0; i < count && rc == MMH_OK; ++i)
    if (hipEventElapsedTime(&ms_each[i], ev[i], ev[i + 1]) != hipSuccess) rc = MMH_ERR_HIP;
  for (auto &e : ev) (void)hipEventDestroy(e);
  return rc;
}

int mmh_probe_mfma_f32(mmh_handle_t h, float *tflops) {
  if (!h || !tflops) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return mmh::probe_mfma_f32(h->cu_count, tflops, &g_last_error);
}

int mmh_probe_mfma_i8(mmh_handle_t h, float *tops) {
  if (!h || !tops) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return mmh::probe_mfma_i8(h->cu_count, tops, &g_last_error);
}

int mmh_probe_mfma_i8_sustained(mmh_handle_t h, int random_operands, float min_ms, float *tops) {
  if (!h || !tops || min_ms < 0.f || min_ms > 2000.f) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return mmh::probe_mfma_i8(h->cu_count, tops, &g_last_error, random_operands ? 1 : 0, min_ms);
}

int mmh_probe_hbm_copy(mmh_handle_t h, size_t bytes, float *gbps) {
  if (!h || !gbps || bytes < (1u << 20)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return mmh::probe_hbm_copy(bytes, gbps, &g_last_error, h->cu_count, 0);
}

int mmh_probe_hbm_read(mmh_handle_t h, size_t bytes, float *gbps) {
  if (!h || !gbps || bytes < (1u << 20)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return mmh::probe_hbm_copy(bytes, gbps, &g_last_error, h->cu_count, 1);
}

#ifdef MMH_DMA_TIMELINE
// timeline build only: where the plain LDS-DMA kernels write their timeline stamps (4 x uint64 per workgroup;
// NULL switches them off).  tools/dma_timeline.py.
int mmh_ab_set_stamps(mmh_handle_t h, void *stamps) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mmh::g_dma_stamps), &stamps, sizeof(void *)));
  return MMH_OK;
}
#endif

int mmh_streamk_plan(long tiles, int nk, int grid, int *order, int *place) {
  if (!order || !place) return MMH_ERR_INVALID_ARG;
  return build_sk_tables(tiles, nk, grid, order, place) ? MMH_OK : MMH_ERR_INVALID_ARG;
}

int mmh_probe_lds_read(mmh_handle_t h, int width, float *gbps) {
  if (!h || !gbps || (width != 16 && width != 8 && width != 4 && width != -8)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return mmh::probe_lds_read(width, gbps, &g_last_error, h->cu_count);
}

}  // extern "C"
