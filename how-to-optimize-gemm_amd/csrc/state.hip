// state.hip -- the handle: life cycle, sticky error, the stream-K / split-K workspaces and who may use
// them when, the phase-order tables of stream-K launches, and mmh_warm.  Host code only.
#include <algorithm>
#include <cstdlib>
#include <new>

#include "internal.hpp"

namespace mmh {

namespace {
thread_local std::string g_last_error;
thread_local std::string g_last_launch;   // which kernel configuration the last sgemm call ran
}  // namespace

void set_last_error(const std::string &s) { g_last_error = s; }
void set_last_launch(const std::string &s) { g_last_launch = s; }
const std::string &last_error_ref() { return g_last_error; }
const std::string &last_launch_ref() { return g_last_launch; }

int hip_fail(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return MMH_ERR_HIP;
}

int DevBuf::reserve(size_t need, std::vector<void *> *retire_to) {
  if (need <= bytes) return MMH_OK;
  if (p) {
    if (retire_to) retire_to->push_back(p);
    else (void)hipFree(p);
  }
  p = nullptr;
  bytes = 0;
  hipError_t e = hipMalloc(&p, need);
  if (e != hipSuccess) {
    set_last_error(std::string("hipMalloc: ") + hipGetErrorString(e));
    return MMH_ERR_ALLOC;
  }
  bytes = need;
  return MMH_OK;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

hipError_t DeviceGuard::enter(int device) {
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) return e;
  if (prev == device) return hipSuccess;
  e = hipSetDevice(device);
  switched = e == hipSuccess;
  return e;
}
DeviceGuard::~DeviceGuard() {
  if (switched) (void)hipSetDevice(prev);
}

int check_sticky(mmh_context *h) {
  if (h && h->sticky && *reinterpret_cast<volatile int *>(h->sticky) != 0) {
    set_last_error("an earlier split-K launch on this handle timed out waiting for its partial tiles: "
                   "its result is invalid (clear with mmh_set_option(h, MMH_OPT_STREAMK_TIMEOUTS, 0))");
    return MMH_ERR_HIP;
  }
  return MMH_OK;
}

int check_gemm_args(int m, int n, int k, const void *A, int lda, const void *B, int ldb, const void *C, int ldc) {
  if (m < 0 || n < 0 || k < 0) return MMH_ERR_INVALID_ARG;
  if (m == 0 || n == 0) return MMH_OK;
  if (!C || ldc < n) return MMH_ERR_INVALID_ARG;
  if (k > 0 && (!A || !B || lda < k || ldb < n)) return MMH_ERR_INVALID_ARG;
  return MMH_OK;
}

bool known_kernel(int kernel) { return mmh_kernel_name(kernel) != nullptr; }

// ---------------------------------------------------------------------------------------------------
// The hand-off workspaces (flags, partial tiles) exist once PER STREAM the handle has launched on.  Launches on
// one stream are ordered by the stream; launches on different streams use different sets, so they may overlap and
// the handle never needs to order one stream behind another -- it keeps no claim on a stream once the call that
// launched on it has returned, and the caller may destroy the stream.  (Rounds 1-2 had ONE set and synchronised
// the previous stream by handle; ROCm 7.2 crashes inside hipEventRecord / hipStreamSynchronize on a destroyed
// stream, it does not return an error.)  Sets are created on first use and kept; beyond eight, the least
// recently used one that no captured graph points at is released after a device-wide synchronisation (its stream
// may be gone).  A launch that is being CAPTURED into a hipGraph uses the capture stream's OWN set (mmh_reserve_stream,
// or an earlier eager launch on that stream, made it), which is then never evicted and only ever retires (never
// frees) a buffer that has to grow; a capture stream without a large-enough set is refused -- no set is ever shared
// between streams.
// A stream must be synchronised before it is destroyed: sets are keyed by the stream's handle value, which the runtime
// may hand out again while the old stream's last launch is still in flight.
// ---------------------------------------------------------------------------------------------------
int workspace_for(mmh_context *ctx, hipStream_t s, long tiles, size_t parts_bytes, int **flags, float **parts) {
  const bool cap = capturing(s);
  mmh_context::StreamWs *w = nullptr;
  for (auto *e : ctx->ws)
    if (e->stream == s) w = e;
  const size_t need_flags = (size_t)tiles * sizeof(int);
  if (cap && (!w || need_flags > w->flags.bytes || parts_bytes > w->parts.bytes || w->flags_dirty)) {
    // Nothing may be allocated (or cleared outside the graph) while a stream is capturing, and a graph must not share
    // hand-off words and partial-tile slots with launches it is not ordered against: a captured stream-K launch uses
    // the CAPTURE STREAM'S OWN set or nothing.  (Round 3 borrowed the most recently used set of any stream -- a replay
    // beside an eager launch on that set's stream then raced on the words, and the caller could not know which stream
    // that was.)  The stream gets a set by launching the shape eagerly once, or from mmh_reserve_stream.
    set_last_error("a stream-K launch cannot allocate its workspaces while the stream is capturing: give the capture stream a "
                   "set of its own first -- mmh_reserve_stream(handle, stream, m, n, k), or one eager call of the shape on "
                   "that stream");
    return MMH_ERR_UNSUPPORTED;
  }
  if (!w) {
    size_t evictable = 0;
    for (auto *e : ctx->ws) evictable += e->captured ? 0 : 1;
    if (evictable >= 8 && !cap) {
      mmh_context::StreamWs *old = nullptr;
      for (auto *e : ctx->ws)
        if (!e->captured && (!old || e->stamp < old->stamp)) old = e;
      HIP_TRY(hipDeviceSynchronize());   // whatever still uses the set -- on a stream that may no longer exist
      old->flags.release();
      old->parts.release();
      old->flags_dirty = true;
      w = old;
    } else {
      w = new (std::nothrow) mmh_context::StreamWs;
      if (!w) return MMH_ERR_ALLOC;
      ctx->ws.push_back(w);
    }
    w->stream = s;
  }
  w->stamp = ++ctx->ws_stamp;
  if (cap) w->captured = true;
  std::vector<void *> *retire = w->captured ? &ctx->retired : nullptr;
  const size_t need = (size_t)tiles * sizeof(int);
  if (need > w->flags.bytes) {
    // (growing a set whose stream still runs a launch: hipFree waits for the device, so freeing is safe eagerly)
    const int rc = w->flags.reserve(std::max(need, (size_t)(256u << 10)), retire);
    if (rc != MMH_OK) return rc;
    w->flags_dirty = true;
  }
  if (w->flags_dirty) {
    // The kernels restore the zeros themselves (the part that finishes a tile resets its word), so the fill runs
    // only when the buffer is new, has grown, or a launch may have died half-way (a sticky error was cleared).
    HIP_TRY(hipMemsetAsync(w->flags.p, 0, w->flags.bytes, s));
    if (!cap) w->flags_dirty = false;   // a fill recorded into a graph does not clean the buffer NOW
  }
  if (parts_bytes > w->parts.bytes) {
    const int rc = w->parts.reserve(std::max(parts_bytes, (size_t)(16u << 20)), retire);
    if (rc != MMH_OK) return rc;
  }
  *flags = static_cast<int *>(w->flags.p);
  *parts = static_cast<float *>(w->parts.p);
  return MMH_OK;
}

void workspaces_suspect(mmh_context *ctx) {
  for (auto *e : ctx->ws) e->flags_dirty = true;
}

// mmh_reserve_stream: `s` gets a workspace set of its own that is large enough for any stream-K launch MMH_KERNEL_AUTO
// (or a forced tile) can make of an m x n x k problem -- eagerly, so that launches of the shape can then be CAPTURED on
// `s`.  Upper bounds: one hand-off word per 64x64 tile; one partial-tile slot per persistent workgroup of the tile
// family with the largest slots x grid (256x256: one per CU; the smaller tiles: 64 KiB per CU between them).
int reserve_stream(mmh_context *ctx, hipStream_t s, int m, int n, int k) {
  if (m <= 0 || n <= 0 || k <= 0) return MMH_ERR_INVALID_ARG;
  if (capturing(s)) {
    set_last_error("mmh_reserve_stream must run before the capture begins (it allocates)");
    return MMH_ERR_UNSUPPORTED;
  }
  const long cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  const long tiles64 = (long)((m + 63) / 64) * ((n + 63) / 64);
  const long tiles256 = (long)((m + 255) / 256) * ((n + 255) / 256);
  size_t parts = (size_t)cus * 3 * 64 * 64 * sizeof(float);                            // 64x64: three workgroups per CU
  parts = std::max(parts, (size_t)cus * 2 * 128 * 64 * sizeof(float));                  // 128x64: two
  parts = std::max(parts, (size_t)cus * 2 * 128 * 128 * sizeof(float));                 // 128x128: one (K2W, K2L) -- two for a forced register-staged tile (64 KiB of LDS); covers its 128x64 tile's three, too
  if (tiles256 >= cus) parts = std::max(parts, (size_t)cus * 256 * 256 * sizeof(float));   // 256x256: one
  int *flags = nullptr;
  float *p = nullptr;
  return workspace_for(ctx, s, tiles64, parts, &flags, &p);
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
// the L2 reuse a plain launch has.  (pure host arithmetic: mmh_streamk_plan exposes it to the CPU tests)
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

// Built on the host at a shape's first launch into PINNED staging memory and uploaded with an asynchronous
// copy on the launch's own stream: no host synchronisation, and a launch that is being captured into a
// hipGraph records the same copy as a node of the graph -- its entry is then pinned in the cache (never
// evicted or rewritten), so a replay finds staging, tables and kernel arguments as they were captured.
int sk_tables_for(mmh_context *ctx, long tiles, int nk, int grid, hipStream_t s, const int **order, const int **place, int min10) {
  *order = *place = nullptr;
  // Worth it from ~1.8 tiles per workgroup (measured): phase order puts the two workgroups that share a
  // tile on different XCDs, so the partial tile crosses the fabric instead of being an L2 hit -- with one
  // tile per workgroup that hand-over is a tenth of the launch (N = 2176 on 128x64 tiles: 139.5 -> 125.0),
  // with two or more the restored L2 reuse wins (N = 3584 on 64x64 tiles: 138.5 -> 146.0).
  // (min10: a tile code's own threshold in tenths of a tile per workgroup.  Round 5, K2W's 128x128 tile: ordered from ONE
  // tile per workgroup -- N = 2560, 400 tiles on 256 workgroups: L2 hit rate 0.40 -> 0.75, 629 -> 227 MB over the fabric
  // per launch (2.9 x the algorithmic bytes), time unchanged, 144.8 / 144.4 TFLOP/s; 2176 .. 2816 within +-0.3 %:
  // profiles/r05_notes.md.  The smaller tiles keep 1.8: 64x64 at N = 1152 loses 3 %, 128x64 at 1536 loses 3 %.)
  if (min10 <= 0 || ctx->sk_order_min10 != 18) min10 = ctx->sk_order_min10;   // (tools build, option 104: one threshold for all)
  if (!ctx->sk_order || tiles > (1L << 20) || tiles * 10 < (long)grid * min10) return MMH_OK;
  const bool cap = capturing(s);
  for (auto *t : ctx->sk_tables)
    if (t->tiles == tiles && t->nk == nk && t->grid == grid && t->uploaded) {
      t->stamp = ++ctx->sk_stamp;
      if (cap) t->pinned = true;   // the entry must outlive the graph
      const bool seen = std::find(t->upload_streams.begin(), t->upload_streams.end(), s) != t->upload_streams.end();
      if (cap || !seen) {
        // a graph must carry its own upload (the eager one is not ordered against the replay), and a launch on a
        // stream that has no upload of these tables in its own order is not ordered behind anybody else's either:
        // upload again, in THIS stream's order (the same bytes to the same place: harmless beside a launch that is
        // reading them).  Streams that have had theirs are remembered -- two streams alternating on one shape upload
        // once each, not once per call (round 3 remembered the last stream only).
        HIP_TRY(hipMemcpyAsync(t->buf.p, t->host, ((size_t)grid + (size_t)tiles) * sizeof(int), hipMemcpyHostToDevice, s));
        if (!cap) {
          if (t->upload_streams.size() >= 16) t->upload_streams.clear();   // (handle values get recycled: keep the list short)
          t->upload_streams.push_back(s);
        }
      }
      *order = static_cast<const int *>(t->buf.p);
      *place = *order + grid;
      return MMH_OK;
    }
  // nothing may be allocated while a stream is capturing: a shape that was never launched eagerly is captured
  // with ranges in plain order (the identity tables are always right)
  if (cap) return MMH_OK;
  // a new shape: a free entry, else the least recently used one that no graph points at; beyond 32 live
  // entries pinned ones only make the cache grow
  mmh_context::SkTable *slot = nullptr;
  size_t unpinned = 0;
  for (auto *t : ctx->sk_tables) {
    unpinned += t->pinned ? 0 : 1;
    // an entry whose build failed (no pinned memory, a shape the builder declines) holds nothing and nobody reads it:
    // it is the first to be reused -- round 3 left such entries in the list, counted against the 32
    if (!t->pinned && !t->uploaded && !slot) slot = t;
  }
  if (slot) {
    // reuse the dead entry as it is
  } else if (unpinned < 32) {
    slot = new (std::nothrow) mmh_context::SkTable;
    if (!slot) return MMH_ERR_ALLOC;
    ctx->sk_tables.push_back(slot);
  } else {
    for (auto *t : ctx->sk_tables)
      if (!t->pinned && (!slot || t->stamp < slot->stamp)) slot = t;
    // the evicted tables may still be read by a launch in flight on any stream
    if (!cap) {
      HIP_TRY(hipDeviceSynchronize());
    } else {
      // nothing may synchronise during capture: take a fresh entry instead
      slot = new (std::nothrow) mmh_context::SkTable;
      if (!slot) return MMH_ERR_ALLOC;
      ctx->sk_tables.push_back(slot);
    }
  }
  const size_t ints = (size_t)grid + (size_t)tiles;
  slot->uploaded = false;
  slot->tiles = 0;
  if (ints > slot->host_ints) {
    if (slot->host) (void)hipHostFree(slot->host);
    slot->host = nullptr;
    slot->host_ints = 0;
    void *hp = nullptr;
    if (hipHostMalloc(&hp, ints * sizeof(int), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return MMH_OK;   // identity tables are always right
    }
    slot->host = static_cast<int *>(hp);
    slot->host_ints = ints;
  }
  if (!build_sk_tables(tiles, nk, grid, slot->host, slot->host + grid)) return MMH_OK;   // identity is always right
  const int rc = slot->buf.reserve(ints * sizeof(int), cap ? &ctx->retired : nullptr);
  if (rc != MMH_OK) return rc;
  HIP_TRY(hipMemcpyAsync(slot->buf.p, slot->host, ints * sizeof(int), hipMemcpyHostToDevice, s));
  slot->host_ints = std::max(slot->host_ints, ints);
  slot->tiles = tiles;
  slot->nk = nk;
  slot->grid = grid;
  slot->stamp = ++ctx->sk_stamp;
  slot->uploaded = true;
  slot->upload_streams.clear();
  if (!cap) slot->upload_streams.push_back(s);
  slot->pinned = cap;
  *order = static_cast<const int *>(slot->buf.p);
  *place = *order + grid;
  return MMH_OK;
}

// ---------------------------------------------------------------------------------------------------
namespace {
bool is_gfx950(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}
}  // namespace

int create_context(mmh_context **out, int device, bool warm) {
  *out = nullptr;
  int count = 0;
  mmh_device_count(&count);
  if (count <= 0 || device < 0 || device >= count) {
    set_last_error("no such HIP device");
    return MMH_ERR_NO_DEVICE;
  }
  if (!is_gfx950(device)) {
    set_last_error("device is not gfx950 (this library carries gfx950 code objects only)");
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
  if (const char *e = std::getenv("MMH_I8_GRID_CAP")) ctx->i8_grid_cap = std::atoi(e) > 0 ? std::atoi(e) : 0;   // test hook (fuzz_i8.py)
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
  (void)hipGetLastError();   // without the word the opt-in split-K launches are simply not used
  {
    void *st = nullptr;
    if (hipMalloc(&st, 64) == hipSuccess && hipMemset(st, 0, 64) == hipSuccess) ctx->sk_stats = static_cast<int *>(st);
    else (void)hipGetLastError();
  }
  *out = ctx;
  // Everything a first launch would otherwise pay for -- code-object load, the > 64 KiB LDS opt-ins, the
  // residency queries, the stream-K workspaces -- happens HERE, where the reference creates its cuBLAS
  // handle (cuda/test_MMult.cpp:43-44), not inside the first timed MY_MMult.  MMH_LAZY=1 defers it to
  // mmh_warm / first use.
  const char *lazy = std::getenv("MMH_LAZY");
  if (warm && !(lazy && *lazy && *lazy != '0')) {
    const int rc = warm_context(ctx);
    if (rc != MMH_OK) {
      destroy_context(ctx);
      *out = nullptr;
      return rc;
    }
  }
  return MMH_OK;
}

int warm_context(mmh_context *h) {
  if (h->warmed) return MMH_OK;
  // scratch: a 128 K-float buffer every kernel family runs ONE tile of itself on (contents irrelevant)
  DevBuf scratch;
  int rc = scratch.reserve((size_t)(1u << 17) * sizeof(float) + (1u << 20));
  if (rc != MMH_OK) return rc;
  HIP_TRY(hipMemsetAsync(scratch.p, 0, scratch.bytes, nullptr));
  float *p = static_cast<float *>(scratch.p);
  if ((rc = warm_reg(h, p, nullptr)) == MMH_OK && (rc = warm_dma(h, p, nullptr)) == MMH_OK &&
      (rc = warm_dma5(h, p, nullptr)) == MMH_OK) rc = warm_valu(h, p, nullptr);
#ifdef MMH_AB_BUILD
  if (rc == MMH_OK) rc = warm_dma32(h, p, nullptr);   // (K2M: tools/ab/)
#endif
  // the hand-off workspaces at the size the largest stream-K launch of a square sweep needs
  if (rc == MMH_OK) {
    // (not fatal: on a device that is short of memory -- beside a torch process, say -- the set is allocated by the
    // first launch that needs one, at the size it needs, as a handle created with MMH_LAZY=1 does)
    int *flags = nullptr;
    float *parts = nullptr;
    if (workspace_for(h, nullptr, 1 << 16, (size_t)(64u << 20), &flags, &parts) != MMH_OK) {   // the null stream's set
      (void)hipGetLastError();
      for (auto *e : h->ws)
        if (e->stream == nullptr) {
          e->flags.release();
          e->parts.release();
          e->flags_dirty = true;
        }
    }
  }
  const hipError_t e = hipStreamSynchronize(nullptr);
  scratch.release();
  if (rc != MMH_OK) return rc;
  if (e != hipSuccess) return hip_fail(e, "mmh_warm: hipStreamSynchronize");
  h->warmed = true;
  set_last_launch("");
  return MMH_OK;
}

void destroy_context(mmh_context *h) {
  if (!h) return;
  DeviceGuard guard;
  (void)guard.enter(h->device);
  (void)hipDeviceSynchronize();
  h->a.release();
  h->b.release();
  h->c.release();
  for (auto *e : h->ws) {
    e->flags.release();
    e->parts.release();
    delete e;
  }
  h->bt.release();
  h->qa.release();
  h->qb.release();
  h->qc.release();
  h->qs.release();
  for (void *p : h->retired) (void)hipFree(p);
  for (auto *t : h->sk_tables) {
    t->buf.release();
    if (t->host) (void)hipHostFree(t->host);
    delete t;
  }
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
  if (h->sk_stats) (void)hipFree(h->sk_stats);
  rocblas_release(h->rocblas);
  hipblaslt_release(h->blaslt);
  delete h;
}

}  // namespace mmh
