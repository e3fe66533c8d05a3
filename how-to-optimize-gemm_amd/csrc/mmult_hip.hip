// mmult_hip.hip -- libmmult_hip.so: the C ABI declared in include/mmult_hip.h
// over the hand-written gfx950 kernels in this directory.
//
// This file is the "thin C-ABI shim" of BASELINE.json's north star: argument
// validation, kernel selection (the reference's `NEW := MMult_xxx` makefile
// switch, cuda/makefile:1-3, made a run-time choice), launch, and the
// host-pointer flavour's staging.  No torch, no CPU fallback: if no gfx950
// device is visible every compute entry point returns MMH_ERR_NO_DEVICE /
// MMH_ERR_HIP.
#include "../../include/mmult_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "igemm_s8.hpp"
#include "probes.hpp"
#include "quant_s8.hpp"
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

}  // namespace

struct mmh_context {
  int device = 0;
  int kernel = MMH_KERNEL_AUTO;
  int cu_count = 0;
  DevBuf a, b, c;          // staging for the host-pointer flavour
  DevBuf bt;               // int8 GEMM: packed (transposed, padded) B
  int igemm_mode = 0;      // 0 auto (packed-B + LDS-DMA), 1 in-kernel transpose, 2 simple
  DevBuf qa, qb, qc, qs;   // quantised GEMM workspace: int8 A, int8 B, int32 C, {amax bits, scales}
  DevBuf flags;            // stream-K per-tile hand-off flags (+1 error word)
  DevBuf parts;            // stream-K partial tiles: one dense BM x BN slot per persistent workgroup
  long flags_tiles = -1;   // where the error word of the last stream-K launch sits
  int streamk = 1;         // allow the persistent stream-K launch for ragged tile counts
  void *rocblas = nullptr; // rocblas_handle, created on first use
};

namespace {

constexpr size_t lds_bytes(int BM, int BN, int KB = mmh::BK) {
  return 2ull * (size_t)KB * (BM + BN) * sizeof(float);
}

template <typename K>
int allow_big_lds(K kernel, size_t bytes) {
  HIP_TRY(mmh::opt_in_big_lds(reinterpret_cast<const void *>(kernel), bytes));
  return MMH_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// SIMPLE: the un-pipelined rung.  SCHED / BUFLD / ABL: see sgemm_mfma.hpp.  The
// buffer-descriptor path needs every byte offset inside a 2 GiB window; larger
// operands fall back to 64-bit global addressing (same kernel, BUFLD = false).
template <int BM, int BN, bool SIMPLE = false, int SCHED = 4, int ABL = 0, bool BUFLD = true, int WTN = 4,
          int WTM = 4, int KB = mmh::BK>
int launch_mfma(int m, int n, int k, const float *A, int lda, const float *B, int ldb,
                float *C, int ldc, int acc, hipStream_t s) {
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  const bool fast = (m % BM == 0) && (n % BN == 0) && (k % KB == 0) && (lda % 4 == 0) &&
                    (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(A) && aligned16(B) &&
                    aligned16(C);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  dim3 grid((unsigned)(nbm * nbn)), block(threads);
  const size_t lim = (1ull << 31) - 4096;
  const bool window_ok = ((size_t)BM * lda + k) * 4 < lim && ((size_t)k * ldb + BN) * 4 < lim;
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
    if (BUFLD && window_ok) MMH_LAUNCH((mmh::sgemm_mfma_kernel<BM, BN, true, SCHED, 0, true, WTN, WTM, KB>));
    else                    MMH_LAUNCH((mmh::sgemm_mfma_kernel<BM, BN, true, SCHED, 0, false, WTN, WTM, KB>));
  } else if (BUFLD && window_ok) {
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

// Persistent chained stream-K launch (sgemm_mfma.hpp, K2p) of tile config
// <BM, BN, WTN>.  Returns MMH_OK if it launched, 1 if the shape does not qualify
// (caller then uses the plain one-tile-per-workgroup launch).
template <int BM, int BN, int WTN, int WTM = 4, int KB = mmh::BK>
int try_launch_streamk(mmh_context *ctx, int m, int n, int k, const float *A, int lda,
                       const float *B, int ldb, float *C, int ldc, int acc, hipStream_t s) {
  if (!ctx || !ctx->streamk) return 1;
  const size_t lim = (1ull << 31) - 4096;
  if (!(((size_t)BM * lda + k) * 4 < lim && ((size_t)k * ldb + BN) * 4 < lim)) return 1;   // descriptor window
  // whole, 16-byte-aligned shapes run the unguarded kernel; everything else the guarded one (partial
  // tiles travel through a workspace, not through C, so C's alignment and ragged edges do not matter)
  const bool fast = (m % BM == 0) && (n % BN == 0) && (k % KB == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                    (ldc % 4 == 0) && aligned16(A) && aligned16(B) && aligned16(C);
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  const long tiles = (long)nbm * nbn;
  const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
  constexpr size_t lds = lds_bytes(BM, BN, KB);
  constexpr int threads = (BM / (16 * WTM)) * (BN / (16 * WTN)) * 64;
  auto kern_fast = mmh::sgemm_mfma_streamk_kernel<BM, BN, false, WTN, WTM, KB>;
  auto kern_edge = mmh::sgemm_mfma_streamk_kernel<BM, BN, true, WTN, WTM, KB>;
  {
    const int ok = allow_big_lds(fast ? kern_fast : kern_edge, lds);
    if (ok != MMH_OK) return ok;
  }
  // resident workgroups per CU: what the runtime reports, never more than LDS allows
  static int per_cu = [&] {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kern_edge, threads, lds) != hipSuccess || v < 1) v = 1;
    const int by_lds = (int)((160 * 1024) / lds);
    return v < by_lds ? v : by_lds;
  }();
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
  int rc = ctx->flags.reserve((size_t)(tiles + 1) * sizeof(int));
  if (rc != MMH_OK) return rc;
  rc = ctx->parts.reserve((size_t)grid * BM * BN * sizeof(float));   // one partial-tile slot per range
  if (rc != MMH_OK) return rc;
  int *flags = static_cast<int *>(ctx->flags.p);
  float *parts = static_cast<float *>(ctx->parts.p);
  ctx->flags_tiles = tiles;
  HIP_TRY(hipMemsetAsync(flags, 0, (size_t)(tiles + 1) * sizeof(int), s));
  if (fast)
    hipLaunchKernelGGL(kern_fast, dim3((unsigned)grid), dim3(threads), lds, s, m, n, k, A, lda, B, ldb, C, ldc,
                       acc, nbm, nbn, flags, flags + tiles, parts);
  else
    hipLaunchKernelGGL(kern_edge, dim3((unsigned)grid), dim3(threads), lds, s, m, n, k, A, lda, B, ldb, C, ldc,
                       acc, nbm, nbn, flags, flags + tiles, parts);
  HIP_TRY(hipGetLastError());
  {
    char buf[176];
    snprintf(buf, sizeof buf,
             "sgemm_mfma_streamk_kernel<%d,%d> wave tile %dx%d, K-slice %d, %s%ld tiles on %d persistent workgroups",
             BM, BN, 16 * WTM, 16 * WTN, KB, fast ? "" : "guarded, ", tiles, grid);
    g_last_launch = buf;
  }
  return MMH_OK;
}

int launch_valu(int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C,
                int ldc, int acc, hipStream_t s) {
  constexpr int BM = 128, BN = 128;
  const int nbm = (m + BM - 1) / BM, nbn = (n + BN - 1) / BN;
  const bool fast = (m % BM == 0) && (n % BN == 0) && (k % mmh::BK == 0) && (lda % 4 == 0) &&
                    (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(A) && aligned16(B) &&
                    aligned16(C);
  constexpr size_t lds = lds_bytes(BM, BN);
  dim3 grid((unsigned)(nbm * nbn)), block(256);
  if (fast)
    hipLaunchKernelGGL(mmh::sgemm_valu_kernel<false>, grid, block, lds, s, m, n, k, A, lda, B, ldb,
                       C, ldc, acc, nbm, nbn);
  else
    hipLaunchKernelGGL(mmh::sgemm_valu_kernel<true>, grid, block, lds, s, m, n, k, A, lda, B, ldb,
                       C, ldc, acc, nbm, nbn);
  HIP_TRY(hipGetLastError());
  g_last_launch = "sgemm_valu_kernel<128,128>";
  return MMH_OK;
}

int launch_naive(int m, int n, int k, const float *A, int lda, const float *B, int ldb, float *C,
                 int ldc, int acc, hipStream_t s) {
  dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 3) / 4)), block(256);
  hipLaunchKernelGGL(mmh::sgemm_naive_kernel, grid, block, 0, s, m, n, k, A, lda, B, ldb, C, ldc,
                     acc);
  HIP_TRY(hipGetLastError());
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
  switch (kernel) {
    case MMH_KERNEL_VALU:
      return launch_valu(m, n, k, dA, lda, dB, ldb, dC, ldc, acc, s);
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
      const long cus = ctx && ctx->cu_count > 0 ? ctx->cu_count : 256;
      const long tiles128 = (long)((m + 127) / 128) * ((n + 127) / 128);
      const long tiles128x64 = (long)((m + 127) / 128) * ((n + 63) / 64);
      const long tiles256 = (long)((m + 255) / 256) * ((n + 255) / 256);
      // At least one 256x256 tile per CU: the big tile (fewest staging ops per MFMA) -- unless its
      // edge tiles pad the shape noticeably more than 128x128 tiles would (an edge tile costs a whole
      // tile's time; ragged tile COUNTS are balanced by stream-K for either size).
      if (tiles256 >= cus) {
        const double fill256 = (double)m * (double)n / ((double)tiles256 * 65536.0);
        const double fill128 = (double)m * (double)n / ((double)tiles128 * 16384.0);
        if (fill256 >= fill128 - 0.015)
          return sgemm_on(ctx, MMH_KERNEL_MFMA_256X256, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate, s);
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
    case 19: {  // A/B: B through LDS-DMA (buffer_load ... lds)
      const int nbm = m / 128, nbn = n / 128;
      if ((m % 128) || (n % 128) || (k % 32) || (lda % 4) || (ldb % 4) || (ldc % 4) || !aligned16(dA) ||
          !aligned16(dB) || !aligned16(dC))
        return MMH_ERR_INVALID_ARG;
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

}  // namespace

// ===========================================================================
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

int mmh_version(void) { return 100; }

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
  *handle = nullptr;
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
  HIP_TRY(hipSetDevice(device));
  mmh_context *ctx = new (std::nothrow) mmh_context;
  if (!ctx) return MMH_ERR_ALLOC;
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  *handle = ctx;
  return MMH_OK;
}

int mmh_destroy(mmh_handle_t h) {
  if (!h) return MMH_OK;
  (void)hipSetDevice(h->device);
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
  mmh::rocblas_release(h->rocblas);
  delete h;
  return MMH_OK;
}

int mmh_set_kernel(mmh_handle_t h, int kernel) {
  if (!h || !mmh_kernel_name(kernel)) return MMH_ERR_INVALID_ARG;
  h->kernel = kernel;
  return MMH_OK;
}

int mmh_set_option(mmh_handle_t h, int option, int value) {
  if (!h) return MMH_ERR_INVALID_ARG;
  if (option == MMH_OPT_STREAMK) {
    h->streamk = value ? 1 : 0;
    return MMH_OK;
  }
  if (option == MMH_OPT_IGEMM_MODE && ((value >= 0 && value <= 6) || (value >= 10 && value <= 13))) {
    h->igemm_mode = value;
    return MMH_OK;
  }
  return MMH_ERR_INVALID_ARG;
}

int mmh_get_option(mmh_handle_t h, int option, int *value) {
  if (!h || !value) return MMH_ERR_INVALID_ARG;
  if (option == MMH_OPT_STREAMK) {
    *value = h->streamk;
    return MMH_OK;
  }
  if (option == MMH_OPT_IGEMM_MODE) {
    *value = h->igemm_mode;
    return MMH_OK;
  }
  if (option == MMH_OPT_STREAMK_TIMEOUTS) {
    *value = 0;
    if (h->flags_tiles < 0 || !h->flags.p) return MMH_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(value, static_cast<int *>(h->flags.p) + h->flags_tiles, sizeof(int),
                      hipMemcpyDeviceToHost));
    return MMH_OK;
  }
  return MMH_ERR_INVALID_ARG;
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
    case MMH_KERNEL_MFMA: return "MMult_hip_mfma";
    case MMH_KERNEL_MFMA_256: return "MMult_hip_mfma256";
    case MMH_KERNEL_NAIVE: return "MMult_hip_naive";
    case MMH_KERNEL_MFMA_SIMPLE: return "MMult_hip_mfma_simple";
    case MMH_KERNEL_MFMA_PIPE: return "MMult_hip_mfma_pipe";
    case MMH_KERNEL_MFMA_TILES: return "MMult_hip_mfma_tiles";
    case MMH_KERNEL_MFMA_128X64: return "MMult_hip_mfma_128x64";
    case MMH_KERNEL_MFMA_64X64: return "MMult_hip_mfma_64x64";
    case MMH_KERNEL_MFMA_256X256: return "MMult_hip_mfma_256x256";
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
    default: return nullptr;
  }
}

int mmh_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
              int ldb, float *dC, int ldc, int accumulate, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  return sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate,
                  static_cast<hipStream_t>(stream));
}

int mmh_sgemm_host(mmh_handle_t h, int m, int n, int k, const float *A, int lda, const float *B,
                   int ldb, float *C, int ldc, int accumulate) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, A, lda, B, ldb, C, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  HIP_TRY(hipSetDevice(h->device));
  // Device images are dense (lda=k, ldb=n, ldc=n) whatever the host strides.
  const size_t ab = (size_t)m * k * sizeof(float), bb = (size_t)k * n * sizeof(float),
               cb = (size_t)m * n * sizeof(float);
  if ((rc = h->a.reserve(ab ? ab : 16)) != MMH_OK) return rc;
  if ((rc = h->b.reserve(bb ? bb : 16)) != MMH_OK) return rc;
  if ((rc = h->c.reserve(cb)) != MMH_OK) return rc;
  float *dA = static_cast<float *>(h->a.p), *dB = static_cast<float *>(h->b.p),
        *dC = static_cast<float *>(h->c.p);
  if (k > 0) {
    HIP_TRY(hipMemcpy2D(dA, (size_t)k * 4, A, (size_t)lda * 4, (size_t)k * 4, m,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy2D(dB, (size_t)n * 4, B, (size_t)ldb * 4, (size_t)n * 4, k,
                        hipMemcpyHostToDevice));
  }
  if (accumulate)
    HIP_TRY(hipMemcpy2D(dC, (size_t)n * 4, C, (size_t)ldc * 4, (size_t)n * 4, m,
                        hipMemcpyHostToDevice));
  rc = sgemm_on(h, h->kernel, m, n, k, dA, k, dB, n, dC, n, accumulate, nullptr);
  if (rc != MMH_OK) return rc;
  HIP_TRY(hipMemcpy2D(C, (size_t)ldc * 4, dC, (size_t)n * 4, (size_t)n * 4, m,
                      hipMemcpyDeviceToHost));
  return MMH_OK;
}

int mmh_igemm_s8(mmh_handle_t h, int m, int n, int k, const int8_t *dA, int lda, const int8_t *dB,
                 int ldb, int32_t *dC, int ldc, int accumulate, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  HIP_TRY(hipSetDevice(h->device));
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
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = h->qs.reserve(64);
  if (rc != MMH_OK) return rc;
  unsigned *amax = static_cast<unsigned *>(h->qs.p) + 8;   // scratch words for stand-alone calls
  HIP_TRY(hipMemsetAsync(amax, 0, 2 * sizeof(unsigned), s));
  const mmh::QuantTensor t{dX, rows, cols, ldx, dQ, ldq}, none{nullptr, 0, 0, 0, nullptr, 0};
  const dim3 g(mmh::quant_rows_grid(rows, 0), 1), gmax(mmh::quant_rows_grid(rows, 0, 512), 1);
  hipLaunchKernelGGL(mmh::absmax_kernel, gmax, dim3(256), 0, s, t, none, mmh::quant_vec_ok(t, false) ? 1 : 0, 0,
                     amax);
  hipLaunchKernelGGL(mmh::quantize_kernel, g, dim3(256), 0, s, t, none, mmh::quant_vec_ok(t, true) ? 1 : 0, 0, amax,
                     d_scale);
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

int mmh_qgemm_f32(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
                  int ldb, float *dC, int ldc, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k == 0) {
    HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * 4, 0, (size_t)n * 4, (size_t)m, s));
    return MMH_OK;
  }
  // dense, 16-byte-friendly workspace images: int8 A (m x ka), int8 B (k x nb), int32 C (m x nb)
  const int ka = (k + 15) & ~15, nb = (n + 3) & ~3;
  if ((rc = h->qa.reserve((size_t)m * ka)) != MMH_OK) return rc;
  if ((rc = h->qb.reserve((size_t)k * nb)) != MMH_OK) return rc;
  if ((rc = h->qs.reserve(64)) != MMH_OK) return rc;
  int8_t *qa = static_cast<int8_t *>(h->qa.p), *qb = static_cast<int8_t *>(h->qb.p);
  unsigned *amax = static_cast<unsigned *>(h->qs.p);        // [0] A, [1] B
  float *scales = reinterpret_cast<float *>(amax + 2);      // [0] A, [1] B
  HIP_TRY(hipMemsetAsync(amax, 0, 2 * sizeof(unsigned), s));
  // A and B share one abs-max launch and one quantisation launch (blockIdx.y picks the tensor)
  const mmh::QuantTensor ta{dA, m, k, lda, qa, ka}, tb{dB, k, n, ldb, qb, nb};
  const dim3 g(mmh::quant_rows_grid(m, k), 2), gmax(mmh::quant_rows_grid(m, k, 512), 2);
  hipLaunchKernelGGL(mmh::absmax_kernel, gmax, dim3(256), 0, s, ta, tb, mmh::quant_vec_ok(ta, false) ? 1 : 0,
                     mmh::quant_vec_ok(tb, false) ? 1 : 0, amax);
  hipLaunchKernelGGL(mmh::quantize_kernel, g, dim3(256), 0, s, ta, tb, mmh::quant_vec_ok(ta, true) ? 1 : 0,
                     mmh::quant_vec_ok(tb, true) ? 1 : 0, amax, scales);
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
  if (m == 0 || n == 0 || k == 0) return sgemm_on(h, MMH_KERNEL_MFMA, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, static_cast<hipStream_t>(stream));
  HIP_TRY(hipSetDevice(h->device));
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

int mmh_sgemm_sharded(int ngpus, int m, int n, int k, const float *A, int lda, const float *B,
                      int ldb, float *C, int ldc, int kernel, float *timings_ms) {
  int rc = check_gemm_args(m, n, k, A, lda, B, ldb, C, ldc);
  if (rc != MMH_OK) return rc;
  if (ngpus <= 0 || !mmh_kernel_name(kernel)) return MMH_ERR_INVALID_ARG;
  int count = 0;
  mmh_device_count(&count);
  if (count < ngpus) {
    g_last_error = "fewer visible devices than ngpus";
    return MMH_ERR_NO_DEVICE;
  }
  return mmh::sgemm_sharded_impl(ngpus, m, n, k, A, lda, B, ldb, C, ldc, kernel, timings_ms,
                                 &g_last_error,
                                 [](int kern, int mm, int nn, int kk, const float *a, int la,
                                    const float *b, int lb, float *c, int lc, hipStream_t s) {
                                   return sgemm_on(nullptr, kern, mm, nn, kk, a, la, b, lb, c, lc, 0, s);
                                 });
}

int mmh_time_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
                   int ldb, float *dC, int ldc, int warmup, int reps, void *stream,
                   float *ms_per_call) {
  if (!h || reps <= 0 || warmup < 0 || !ms_per_call) return MMH_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
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
  return MMH_OK;
}

int mmh_probe_mfma_f32(mmh_handle_t h, float *tflops) {
  if (!h || !tflops) return MMH_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  return mmh::probe_mfma_f32(h->cu_count, tflops, &g_last_error);
}

int mmh_probe_mfma_i8(mmh_handle_t h, float *tops) {
  if (!h || !tops) return MMH_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  return mmh::probe_mfma_i8(h->cu_count, tops, &g_last_error);
}

int mmh_probe_mfma_i8_sustained(mmh_handle_t h, int random_operands, float min_ms, float *tops) {
  if (!h || !tops || min_ms < 0.f || min_ms > 2000.f) return MMH_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  return mmh::probe_mfma_i8(h->cu_count, tops, &g_last_error, random_operands ? 1 : 0, min_ms);
}

int mmh_probe_hbm_copy(mmh_handle_t h, size_t bytes, float *gbps) {
  if (!h || !gbps || bytes < (1u << 20)) return MMH_ERR_INVALID_ARG;
  HIP_TRY(hipSetDevice(h->device));
  return mmh::probe_hbm_copy(bytes, gbps, &g_last_error);
}

}  // extern "C"
