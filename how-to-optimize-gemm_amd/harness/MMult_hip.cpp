// MMult_hip.cpp -- the ONE symbol a reference-shaped harness links against:
//
//   void MY_MMult(int m, int n, int k, float *a, int lda, float *b, int ldb,
//                 float *c, int ldc)            (_Z8MY_MMultiiiPfiS_iS_i)
//
// host-pointer flavour, C = A*B + C, exactly the contract of
// armv7/MMult0.c:9-24 / aarch64/MMult0.cpp:3-19 as called from
// armv7/test_MMult.c:76 and aarch64/test_MMult.cpp:113.  It forwards to the C
// ABI (mmh_sgemm_host); all arithmetic happens in the gfx950 kernels.
//
// A second overload mirrors the CUDA directory's device-pointer flavour
// (cuda/test_MMult.cpp:13-14: leading handle argument, C = A*B, asynchronous); a third the
// vulkan directory's (vulkan/test_MMult.cpp:10,55: six arguments, host pointers, dense row-major,
// C = A*B, returns the device milliseconds of the GEMM -- _Z8MY_MMultiiiPfS_S_).
//
// Kernel variant: environment variable MMULT_KERNEL = auto (default) | mfma | mfma256 |
// mfma_256x256 | mfma_128x64 | mfma_64x64 | mfma_*_dma | mfma_pipe | mfma_simple | valu | naive ... -- any short name
// mmh_kernel_id knows (the run-time form of the reference's `NEW := MMult_xxx`, cuda/makefile:3).  MMULT_HOST_PANELS = -1 (default, automatic) | 0 (plain staged
// form) | 2..16: row panels of mmh_sgemm_host's copy/compute pipeline.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/mmult_hip.h"

namespace {

int kernel_from_env() {
  const char *e = std::getenv("MMULT_KERNEL");
  if (!e || !*e) return MMH_KERNEL_AUTO;
  const int id = mmh_kernel_id(e);   // the library's own table of short names
  if (id >= 0) return id;
  std::fprintf(stderr, "MMULT_KERNEL=%s is not a kernel variant\n", e);
  std::exit(EXIT_FAILURE);
}

void die(int st, const char *what) {
  std::fprintf(stderr, "MY_MMult: %s failed: %s (%s)\n", what, mmh_strerror(st), mmh_last_error());
  std::exit(EXIT_FAILURE);   // the reference's checkCudaErrors behaviour (cuda/helper.h:10-14)
}

mmh_handle_t default_handle() {
  static mmh_handle_t h = [] {
    mmh_handle_t hh = nullptr;
    int dev = 0;
    if (const char *e = std::getenv("MMULT_DEVICE")) dev = std::atoi(e);
    int st = mmh_create(&hh, dev);
    if (st != MMH_OK) die(st, "mmh_create");
    st = mmh_set_kernel(hh, kernel_from_env());
    if (st != MMH_OK) die(st, "mmh_set_kernel");
    if (const char *e = std::getenv("MMULT_HOST_PANELS")) {   // row panels of the copy/compute pipeline
      st = mmh_set_option(hh, MMH_OPT_HOST_PANELS, std::atoi(e));
      if (st != MMH_OK) die(st, "mmh_set_option(MMH_OPT_HOST_PANELS)");
    }
    return hh;
  }();
  return h;
}

}  // namespace

// host flavour: C += A*B
void MY_MMult(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c, int ldc) {
  const int st = mmh_sgemm_host(default_handle(), m, n, k, a, lda, b, ldb, c, ldc, /*accumulate=*/1);
  if (st != MMH_OK) die(st, "mmh_sgemm_host");
}

// device flavour: C = A*B on device pointers, enqueued on the null stream, no sync
void MY_MMult(mmh_handle_t handle, int m, int n, int k, float *d_A, int lda, float *d_B, int ldb,
              float *d_C, int ldc) {
  const int st = mmh_sgemm(handle, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, /*accumulate=*/0, nullptr);
  if (st != MMH_OK) die(st, "mmh_sgemm");
}

// vulkan-directory flavour: C = A*B on dense row-major host buffers; returns the GEMM's device time in ms
float MY_MMult(int m, int n, int k, float *a, float *b, float *c) {
  float ms = 0.0f;
  const int st = mmh_sgemm_host_timed(default_handle(), m, n, k, a, k, b, n, c, n, /*accumulate=*/0, &ms);
  if (st != MMH_OK) die(st, "mmh_sgemm_host_timed");
  return ms;
}
