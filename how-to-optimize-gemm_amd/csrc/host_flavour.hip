// host_flavour.hip -- the host-pointer flavour of MY_MMult (armv7/test_MMult.c:8,76; aarch64/test_MMult.cpp:17,113):
// mmh_sgemm_host stages A, B (and C) through device buffers the handle owns; large problems run as a row-panel
// copy / compute pipeline.  mmh_sgemm_host_timed is the vulkan directory's flavour (returns the GEMM's device time).
// Part of libmmult_hip.so (see internal.hpp).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "internal.hpp"

namespace mmh {
namespace {

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
  hipError_t e = hipStreamCreateWithFlags(&h->hs_in, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->hs_run, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->hs_out, hipStreamNonBlocking);
  for (int i = 0; i < kMaxHostPanels && e == hipSuccess; ++i) {
    e = hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_run[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming);
  if (e != hipSuccess) {
    // a partial set is of no use and would be created again (and leaked) by the next call: give back what exists
    for (int i = 0; i < kMaxHostPanels; ++i) {
      if (h->ev_in[i]) { (void)hipEventDestroy(h->ev_in[i]); h->ev_in[i] = nullptr; }
      if (h->ev_run[i]) { (void)hipEventDestroy(h->ev_run[i]); h->ev_run[i] = nullptr; }
    }
    if (h->ev_b) { (void)hipEventDestroy(h->ev_b); h->ev_b = nullptr; }
    if (h->hs_in) { (void)hipStreamDestroy(h->hs_in); h->hs_in = nullptr; }
    if (h->hs_run) { (void)hipStreamDestroy(h->hs_run); h->hs_run = nullptr; }
    if (h->hs_out) { (void)hipStreamDestroy(h->hs_out); h->hs_out = nullptr; }
    return hip_fail(e, "host flavour: creating the copy / compute streams and their events");
  }
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
      set_last_error(out_err);
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
}  // namespace mmh

using namespace mmh;

extern "C" {

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

}  // extern "C"
