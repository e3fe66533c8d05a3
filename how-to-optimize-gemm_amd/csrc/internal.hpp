// internal.hpp -- what the translation units of libmmult_hip.so share on the HOST side: the handle,
// error plumbing, workspace ownership and the launcher entry points each TU exports to the others.
//
// The library is the "thin C-ABI shim" of BASELINE.json's north star around the hand-written gfx950
// kernels of this directory.  Translation units (build.py compiles them in parallel):
//   state.hip        handle life cycle, sticky error, stream-K workspaces and phase tables, mmh_warm
//   policy.hip       sgemm_on(): MMH_KERNEL_AUTO's tile choice -- the reference's `NEW := MMult_xxx`
//                    makefile switch (cuda/makefile:1-3) made a run-time choice.  Pure host code.
//   launch_reg.hip   register-staged MFMA tiles (sgemm_mfma.hpp): plain, stream-K, split-K
//   launch_dma.hip   LDS-DMA tiles (sgemm_dma.hpp): plain, stream-K; whole and guarded shapes
//   launch_dma5.hip  LDS-DMA tiles with a loader wave (sgemm_dma5.hpp): plain, chained stream-K
//   launch_valu.hip  K1 / K0 (sgemm_valu.hpp)
//   host_flavour.hip mmh_sgemm_host(_timed): the host-pointer MY_MMult, row-panel pipeline
//   shard.hip        mmh_shard_*: single-process row-panel shard over RCCL
//   igemm.hip        int8 GEMM, quantisation passes
//   vendor.hip       rocBLAS / hipBLASLt comparators, RCCL loader
//   probes.hip       peak probes
//   abi.hip          the remaining extern "C" entry points
// No torch, no CPU fallback: without a gfx950 device every compute entry point returns
// MMH_ERR_NO_DEVICE / MMH_ERR_HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "ab_build.hpp"
#include "../../include/mmult_hip.h"

// Kernel ids of the tools build (libmmult_hip_ab.so) that name whole tile families; the product library neither
// defines nor accepts them.  K2M (tools/ab/sgemm_dma32.hpp, round 4): the LDS-DMA ring feeding v_mfma_f32_32x32x2_f32 -- 64-cycle
// matrix instructions, one conflict-free ds_read_b128 + two v_permlane32_swap per eight k's of A -- and the two-block
// form v_mfma_f32_32x32x1_2b_f32; measured slower than the 16x16x4 tiles (profiles/r04_notes.md).
#define MMH_KERNEL_MFMA32_64X64_DMA 48
#define MMH_KERNEL_MFMA32_128X64_DMA 49
#define MMH_KERNEL_MFMA32_128X128_DMA 50
#define MMH_KERNEL_MFMA32_64X128_DMA 51
#define MMH_KERNEL_MFMA32B_128X64_DMA 60
#define MMH_KERNEL_MFMA32B_64X128_DMA 61
#define MMH_KERNEL_MFMA32B_128X128_DMA 62

namespace mmh {

// ---- error text (thread-local, state.hip) ----
void set_last_error(const std::string &s);
void set_last_launch(const std::string &s);
const std::string &last_error_ref();
const std::string &last_launch_ref();
int hip_fail(hipError_t e, const char *what);

#define HIP_TRY(expr)                                        \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return ::mmh::hip_fail(e_, #expr); \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  // keep_old: a launch captured into a hipGraph may point at the current allocation -- then a buffer
  // that has to grow is RETIRED (freed with the handle), never freed under the graph
  int reserve(size_t need, std::vector<void *> *retire_to = nullptr);
  void release();
};

// Entry points run on the HANDLE's device and leave the caller's current device as they found it
// (a torch process whose current device differs from the handle's must not find it changed).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t enter(int device);
  ~DeviceGuard();
};

constexpr int kMaxHostPanels = 16;

struct GemmArgs {
  int m, n, k;
  const float *A;
  int lda;
  const float *B;
  int ldb;
  float *C;
  int ldc;
  int acc;          // 0: C = A*B, 1: C = A*B + C
  hipStream_t s;
  // "rim" launches (sgemm_dma.hpp): m x n above is the TRIMMED problem the tiles cover, rim_m x rim_n the whole
  // one -- the strips in between run on the vector ALU in extra workgroups of the same launch.  0: no rim.
  int rim_m = 0, rim_n = 0;
  // launch form: 0 = the launcher's own rule (a kernel the caller forced), 1 = one workgroup per tile, 2 = the
  // persistent stream-K launch -- what MMH_KERNEL_AUTO's cost table decided (policy.hip)
  int form = 0;
  // ... and the persistent workgroups per CU the table priced that launch on (0: whatever the kernel's residency allows)
  int sk_w = 0;
};

}  // namespace mmh

struct mmh_context {
  int device = 0;
  int kernel = MMH_KERNEL_AUTO;
  int cu_count = 0;
  mmh::DevBuf a, b, c;     // staging for the host-pointer flavour
  mmh::DevBuf bt;          // int8 GEMM: packed (transposed, padded) B
  int igemm_mode = 0;      // 0 auto, see MMH_OPT_IGEMM_MODE
  int i8_grid_cap = 0;     // test hook (environment MMH_I8_GRID_CAP, read at mmh_create): K3p's persistent grid, so that small shapes walk several tiles per workgroup
  mmh::DevBuf qa, qb, qc, qs;   // quantised GEMM workspace: int8 A, int8 B, int32 C, {amax bits, scales}
  // stream-K / split-K workspaces, one set PER STREAM the handle has launched on: launches on different streams
  // never share hand-off words or partial tiles, so nothing has to order one stream behind another and the handle
  // never touches a stream again after the call that used it returns (the caller may destroy it).
  struct StreamWs {
    hipStream_t stream = nullptr;
    mmh::DevBuf flags;     // per-tile hand-off words; every launch leaves them ZERO (the last reader of a word
                           // resets it), so only a fresh or suspect buffer is memset
    bool flags_dirty = true;
    mmh::DevBuf parts;     // partial tiles
    unsigned long stamp = 0;
    bool captured = false; // a captured launch points at flags / parts: retire on growth, never evict
  };
  std::vector<StreamWs *> ws;
  unsigned long ws_stamp = 0;
  int *sk_stats = nullptr;           // device: [0] stream-K hand-overs finished by the head's owner (diagnostic)
  std::vector<void *> retired;       // allocations a captured graph may still point at
  int streamk = 1;         // allow the persistent stream-K launch for ragged tile counts
  int splitk = 0;          // opt-in split-K: 0 off (default), 1 auto, >= 2 that many parts
  int host_panels = -1;    // host flavour: -1 auto, 0/1 the plain staged form, n pipelined row panels
  void *rocblas = nullptr; // rocblas_handle, created on first use
  void *blaslt = nullptr;  // hipBLASLt bridge state (vendor.hip)
  // the sticky error word: host memory the device can write (an opt-in split-K wait that times out adds
  // to it); every entry point looks at it before doing anything else
  int *sticky = nullptr;       // host view
  int *sticky_dev = nullptr;   // device view of the same word
  long long spin_limit = 1ll << 26;
  int fault = 0;               // MMH_OPT_FAULT_INJECT
  int pin = 1;                 // persistent launches ask for 160 KiB / w of LDS so that exactly w workgroups fit a CU
  int sk_order = 1;            // stream-K launches get the phase-ordered range / tile tables
  int sk_order_min10 = 18;     // ... from this many tiles per workgroup, in tenths (tools build: option 104)
  int dma_edge = 1;            // ragged / 4-byte-aligned shapes may run the guarded LDS-DMA tiles (MMH_OPT_DMA_EDGE)
  int dma_dword_rows = 1;      // ... including operands whose rows are only 4-byte aligned (odd lda / ldb / base)
  int sk_chain = 1;            // stream-K launches of the K2W tiles (launch_dma5.hip) run a range's parts as one stream of slices (MMH_OPT_STREAMK_CHAIN)
  int rim5 = 0;                // tools build only (MMH_OPT_RIM5): the RIM launch of the 64x64 K2W tile -- measured, it loses
  int ab_whole_ranges = 0;     // tools build only (option 106): persistent launches take WHOLE tiles (ranges rounded to tile boundaries: no hand-over)
  int ab_nodefer = 0;          // tools build only (option 102): stream-K heads publish on the spot (no deferred publish)
  int ab_valu_old = 0;         // tools build only (option 105): the K1 ids run the register-staged K1 of rounds 1-4, not K1W
  int ab_own_occ = 0;          // tools build only (option 103): a whole-tile stream-K launch is bounded by ITS OWN instantiation's residency
  int split_tail = 1;          // a plain K2W launch whose last round the dispatcher would pack two per CU goes out as two launches (launch_dma5.hip; tools build: option 107 switches it off)
  int ab_group_m = 0;          // tools build only (option 101): raster group height of the plain K2W launch, 0 = GROUP_M
  int persist = 0;             // whole rounds of the persistent grid run persistent too (MMH_OPT_PERSIST)
  int rim = 0;                 // MMH_KERNEL_AUTO trims up to this many rows / columns past a 64-boundary off the tiles (MMH_OPT_RIM; off: measured, it does not pay)
  // stream-K tables per launch shape (tiles, K-slices, grid): [order: grid ints][place: tiles ints]
  struct SkTable {
    long tiles = 0;
    int nk = 0, grid = 0;
    mmh::DevBuf buf;
    int *host = nullptr;       // pinned staging the asynchronous upload reads from
    size_t host_ints = 0;
    unsigned long stamp = 0;
    bool pinned = false;       // a captured graph points at buf: never evicted
    bool uploaded = false;
    std::vector<hipStream_t> upload_streams;   // the streams in whose order an upload of these bytes already sits
  };
  std::vector<SkTable *> sk_tables;
  unsigned long sk_stamp = 0;
  // resident workgroups per CU of each persistent kernel, per handle (= per device)
  std::vector<std::pair<const void *, int>> per_cu;
  bool warmed = false;
  // host flavour pipeline: copy-in / compute / copy-out streams, per-panel events
  hipStream_t hs_in = nullptr, hs_run = nullptr, hs_out = nullptr;
  hipEvent_t ev_in[mmh::kMaxHostPanels] = {}, ev_run[mmh::kMaxHostPanels] = {}, ev_b = nullptr;
  bool pipeline_ready = false;
  hipEvent_t t0 = nullptr, t1 = nullptr;   // mmh_sgemm_host_timed
};

namespace mmh {

// every compute entry point: refuse a handle whose sticky error word is set
int check_sticky(mmh_context *h);
#define ENTER(h)                                       \
  ::mmh::DeviceGuard guard_;                           \
  HIP_TRY(guard_.enter((h)->device));                  \
  if (int st_ = ::mmh::check_sticky(h); st_ != MMH_OK) return st_

int create_context(mmh_context **out, int device, bool warm = true);   // warm: unless the environment says MMH_LAZY=1
void destroy_context(mmh_context *h);
int warm_context(mmh_context *h);

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool capturing(hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  return cap != hipStreamCaptureStatusNone;
}

// whole tiles, 16-byte aligned operands: the unguarded instantiations
inline bool fast_shape(int BM, int BN, int KB, const GemmArgs &g) {
  return (g.m % BM == 0) && (g.n % BN == 0) && (g.k % KB == 0) && (g.lda % 4 == 0) && (g.ldb % 4 == 0) &&
         (g.ldc % 4 == 0) && aligned16(g.A) && aligned16(g.B) && aligned16(g.C);
}
// the buffer-descriptor path needs every byte offset inside a 2 GiB window
inline bool window_ok(int BM, int BN, int k, int lda, int ldb) {
  const size_t lim = (1ull << 31) - 4096;
  return ((size_t)BM * lda + k) * 4 < lim && ((size_t)k * ldb + BN) * 4 < lim;
}

int check_gemm_args(int m, int n, int k, const void *A, int lda, const void *B, int ldb, const void *C, int ldc);
bool known_kernel(int kernel);

// ---- stream-K workspaces (state.hip) ----
// the stream's own hand-off words (>= tiles of them, all zero) and partial-tile slots (>= parts_bytes)
int workspace_for(mmh_context *ctx, hipStream_t s, long tiles, size_t parts_bytes, int **flags, float **parts);
int reserve_stream(mmh_context *ctx, hipStream_t s, int m, int n, int k);   // mmh_reserve_stream
void workspaces_suspect(mmh_context *ctx);   // a launch may have died half-way: every set is memset before its next use
bool build_sk_tables(long tiles, int nk, int grid, int *order, int *place);
int sk_tables_for(mmh_context *ctx, long tiles, int nk, int grid, hipStream_t s, const int **order, const int **place, int min10 = 0);

// ---- the kernel families (each returns MMH_OK, an error, or 1 = "this shape does not qualify") ----
int auto_plan(int m, int n, int k, int lda, int ldb, int ldc, int base_align, int cu_count, int *kernel, long *tiles,
              int *streamk_grid);   // policy.hip: mmh_auto_plan
int sgemm_on(mmh_context *ctx, int kernel, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
             float *dC, int ldc, int accumulate, hipStream_t s);
// launch_reg.hip: `kernel` is one of the register-staged ids (MFMA, MFMA_TILES, MFMA_256, MFMA_256X256, MFMA_128X64,
// MFMA_64X64, MFMA_SIMPLE, MFMA_PIPE, the split-K ids and, in the A/B build, the ablation ids)
int launch_reg(mmh_context *ctx, int kernel, const GemmArgs &g);
int launch_reg_splitk(mmh_context *ctx, int tile /* 128 or 64 = BN */, int S, const GemmArgs &g);
int warm_reg(mmh_context *ctx, float *scratch, hipStream_t s);
// launch_dma.hip: tile = MMH_KERNEL_MFMA_{64X64,128X64,128X128}_DMA; returns 1 when the shape does not qualify
int launch_dma(mmh_context *ctx, int kernel, const GemmArgs &g);
bool dma_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g);
int warm_dma(mmh_context *ctx, float *scratch, hipStream_t s);
// tools/ab/launch_dma32.hip (tools build only): tile = MMH_KERNEL_MFMA32_*_DMA (tools/ab/sgemm_dma32.hpp); returns 1 when the shape does not qualify
int launch_dma32(mmh_context *ctx, int kernel, const GemmArgs &g);
bool dma32_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g);
int warm_dma32(mmh_context *ctx, float *scratch, hipStream_t s);
// The RIM launch of the 64x64 K2W tile (sgemm_dma5.hpp, rim_wave): m and / or n ONE element past a multiple of 64, at
// least one whole tile each way.  *r_m / *r_n: rim rows / columns (0 or 1).
// Which plain K2W launches go out as two (launch_dma5.hip, "the tail split"): one whole round of w workgroups per CU and a
// last round of just under one tile per CU, deep enough in K.  Shared with the cost table (policy.hip).
inline bool dma5_tail_split(long tiles, long w, long cus, int k) {
  const long rem = tiles - w * cus;
  return w >= 2 && k >= 512 && 100 * rem > 85 * cus && rem <= cus;
}

inline bool dma5_rim_dims(int m, int n, int *r_m, int *r_n) {
  const int rm = m % 64, rn = n % 64;
  const int a = (rm == 1 && m > 64) ? 1 : 0, b = (rn == 1 && n > 64) ? 1 : 0;
  if (r_m) *r_m = a;
  if (r_n) *r_n = b;
  return a > 0 || b > 0;
}
// launch_dma5.hip: tile = MMH_KERNEL_MFMA_*_DMA5 (sgemm_dma5.hpp); returns 1 when the shape does not qualify
int launch_dma5(mmh_context *ctx, int kernel, const GemmArgs &g);
bool dma5_shape_ok(const mmh_context *ctx, int kernel, const GemmArgs &g);
int warm_dma5(mmh_context *ctx, float *scratch, hipStream_t s);
// launch_valu.hip
int launch_valu(mmh_context *ctx, int kernel, const GemmArgs &g);
int warm_valu(mmh_context *ctx, float *scratch, hipStream_t s);

// ---- vendor bridges (vendor.hip) ----
int rocblas_sgemm_rowmajor(void **handle, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                           float *dC, int ldc, void *stream);
void rocblas_release(void *&handle);
int hipblaslt_sgemm_rowmajor(void **state, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                             float *dC, int ldc, void *stream);
void hipblaslt_release(void *&state);

struct RcclApi {
  void *lib = nullptr;
  int (*get_version)(int *) = nullptr;
  int (*comm_init_all)(void **, int, const int *) = nullptr;
  int (*comm_destroy)(void *) = nullptr;
  int (*comm_count)(void *, int *) = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  int (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  bool ok = false;
};
RcclApi &rccl_api();

}  // namespace mmh
