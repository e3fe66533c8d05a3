// abi.hip -- the extern "C" entry points of include/mmult_hip.h that are not tied to one kernel family: library
// and handle queries, options, the device-pointer MY_MMult (mmh_sgemm), the measurement helpers.
// Part of libmmult_hip.so (see internal.hpp).
#include <algorithm>

#include "internal.hpp"

using namespace mmh;

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

const char *mmh_last_error(void) { return last_error_ref().c_str(); }

const char *mmh_last_launch(void) { return last_launch_ref().c_str(); }

int mmh_version(void) { return 300; }

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
      workspaces_suspect(h);   // a launch that timed out may have left hand-off counters behind
      return MMH_OK;
    case MMH_OPT_IGEMM_MODE:
      if ((value >= 0 && value <= 9 && value != 1 && value != 3 && value != 4)
#ifdef MMH_AB_BUILD
          || value == 1 || value == 3 || value == 4 || (value >= 10 && value <= 13)   // tools/ab/igemm_s8_k3.hpp
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
      workspaces_suspect(h);
      return MMH_OK;
    case MMH_OPT_STREAMK_ORDER:
      h->sk_order = value ? 1 : 0;
      return MMH_OK;
    case MMH_OPT_STREAMK_DELEGATIONS:   // writing 0 resets the counter
      if (value != 0) return MMH_ERR_INVALID_ARG;
      if (h->sk_stats) {
        DeviceGuard guard;
        HIP_TRY(guard.enter(h->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(h->sk_stats, 0, 64));
      }
      return MMH_OK;
    case MMH_OPT_DMA_EDGE:   // 0: ragged / unaligned shapes on the register-staged tiles only; 1: guarded LDS-DMA tiles
      h->dma_edge = value ? 1 : 0;     //    for rows that are 16-byte aligned; 2 (default): for any 4-byte aligned rows
      h->dma_dword_rows = value >= 2 ? 1 : 0;
      return MMH_OK;
    case MMH_OPT_RIM:   // the rim lives in the tools build (measured: it does not pay); the product accepts "off" only
#ifdef MMH_AB_BUILD
      if (value < 0 || value > 16) return MMH_ERR_INVALID_ARG;
#else
      if (value != 0) return MMH_ERR_INVALID_ARG;
#endif
      h->rim = value;
      return MMH_OK;
    case MMH_OPT_STREAMK_CHAIN:
      h->sk_chain = value ? 1 : 0;
      return MMH_OK;
    case MMH_OPT_PERSIST:
      if (value < 0 || value > 1) return MMH_ERR_INVALID_ARG;
      h->persist = value;
      return MMH_OK;
    case MMH_OPT_RIM5:   // (tools build: the fused rim; the product accepts "off" only)
#ifndef MMH_AB_BUILD
      if (value != 0) return MMH_ERR_INVALID_ARG;
#endif
      h->rim5 = value ? 1 : 0;
      return MMH_OK;
#ifdef MMH_AB_BUILD
    case 100:   // A/B: pin the residency of persistent launches by their LDS request (default on)
      h->pin = value ? 1 : 0;
      return MMH_OK;
    case 101:   // A/B: raster group height of the plain K2W launch (0 = the product's GROUP_M)
      if (value < 0 || value > 1024) return MMH_ERR_INVALID_ARG;
      h->ab_group_m = value;
      return MMH_OK;
    case 102:   // A/B: chained stream-K heads publish on the spot instead of on the next part's first slice
      h->ab_nodefer = value ? 1 : 0;
      return MMH_OK;
    case 103:   // A/B (prepared at the end of round 4, not yet measured): whole-tile stream-K launches of the K2W tiles bounded by
                // their own instantiation's residency (77 / 117 registers: three / two workgroups per CU) instead of the guarded one's
      h->ab_own_occ = value ? 1 : 0;
      return MMH_OK;
    case 106:   // A/B (round 6): persistent launches of ragged counts with WHOLE-tile ranges -- no partial tiles, no hand-over, a
                // deterministic share per CU where a plain launch's last round is placed greedily (profiles/r06_notes.md section 6)
      h->ab_whole_ranges = value ? 1 : 0;
      return MMH_OK;
    case 107:   // A/B (round 6): the tail split of plain K2W launches (launch_dma5.hip) on (product) / off
      h->split_tail = value ? 1 : 0;
      return MMH_OK;
    case 105:   // A/B: the vector-ALU rung as it was before round 5 (register-staged K1) instead of K1W
      h->ab_valu_old = value ? 1 : 0;
      return MMH_OK;
    case 104:   // A/B: phase-ordered stream-K tables from this many tiles per workgroup, in tenths (product: 18)
      if (value < 10 || value > 1000) return MMH_ERR_INVALID_ARG;
      h->sk_order_min10 = value;
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
    case MMH_OPT_STREAMK_DELEGATIONS: {
      *value = 0;
      if (!h->sk_stats) return MMH_OK;
      DeviceGuard guard;
      HIP_TRY(guard.enter(h->device));
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(value, h->sk_stats, sizeof(int), hipMemcpyDeviceToHost));
      return MMH_OK;
    }
    case MMH_OPT_DMA_EDGE: *value = h->dma_edge ? (h->dma_dword_rows ? 2 : 1) : 0; return MMH_OK;
    case MMH_OPT_RIM: *value = h->rim; return MMH_OK;
    case MMH_OPT_STREAMK_CHAIN: *value = h->sk_chain; return MMH_OK;
    case MMH_OPT_PERSIST: *value = h->persist; return MMH_OK;
    case MMH_OPT_RIM5: *value = h->rim5; return MMH_OK;
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
    case MMH_KERNEL_VALU_128X64: return "MMult_hip_valu_128x64";
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
    case MMH_KERNEL_MFMA_64X64_DMA5: return "MMult_hip_mfma_64x64_dma5";
    case MMH_KERNEL_MFMA_128X64_DMA5: return "MMult_hip_mfma_128x64_dma5";
    case MMH_KERNEL_MFMA_128X128_DMA5: return "MMult_hip_mfma_128x128_dma5";
    case MMH_KERNEL_MFMA_96X96_DMA5: return "MMult_hip_mfma_96x96_dma5";
    case MMH_KERNEL_MFMA_96X64_DMA5: return "MMult_hip_mfma_96x64_dma5";
    case MMH_KERNEL_MFMA_160X160_DMA5: return "MMult_hip_mfma_160x160_dma5";
    case MMH_KERNEL_MFMA_SPLITK: return "MMult_hip_mfma_splitk";
    case MMH_KERNEL_MFMA_SPLITK_128X64: return "MMult_hip_mfma_splitk_128x64";
#ifdef MMH_AB_BUILD
    case MMH_KERNEL_MFMA32_64X64_DMA: return "MMult_hip_mfma32_64x64_dma";
    case MMH_KERNEL_MFMA32_128X64_DMA: return "MMult_hip_mfma32_128x64_dma";
    case MMH_KERNEL_MFMA32_64X128_DMA: return "MMult_hip_mfma32_64x128_dma";
    case MMH_KERNEL_MFMA32_128X128_DMA: return "MMult_hip_mfma32_128x128_dma";
    case MMH_KERNEL_MFMA32B_128X64_DMA: return "MMult_hip_mfma32b_128x64_dma";
    case MMH_KERNEL_MFMA32B_64X128_DMA: return "MMult_hip_mfma32b_64x128_dma";
    case MMH_KERNEL_MFMA32B_128X128_DMA: return "MMult_hip_mfma32b_128x128_dma";
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
    case 52: return "abl32_128x64_no_swap";
    case 53: return "abl32_128x64_no_dma";
    case 54: return "abl32_128x64_no_a_reads";
    case 55: return "abl32_128x64_mfma_only";
    case 56: return "abl32_64x64_no_swap";
    case 57: return "abl32_64x64_no_dma";
    case 58: return "abl32_64x64_no_a_reads";
    case 59: return "abl32_64x64_mfma_only";
    case 64: return "exp5_64x64_l1d2";
    case 65: return "exp5_64x64_ring6";
    case 66: return "exp5_64x64_ring4";
    case 67: return "exp5_64x64_ring6_l4";
    case 68: return "exp5_128x64_l1d2";
    case 69: return "exp5_128x64_ring4";
    case 72: return "exp5_128x128_l1d2";
    case 79: return "exp5_160x96_l1d2";
    case 80: return "exp5_160x160_l1d2";
    case 82: return "exp5_160x160_l2";
    case 83: return "exp5_96x64_l4";
    case 84: return "exp5_96x64_l2";
    case 85: return "exp5_64x96_l2";
    case 87: return "k1w_128x128_b3l2a4";
    case 89: return "k1w_64x64_a2";
    case 91: return "k1w_64x128";
    case 92: return "k1w_64x64_a4";
    case 93: return "k1w_128x128_b2l4a2";
    case 94: return "k1w_128x128_b2l1a2";
    case 95: return "exp5_160x160_rs0";
    case 96: return "exp5_128x128_rs0";
    case 97: return "exp5_128x64_rs0";
    case 98: return "exp5_64x64_rs0";
    case 99: return "exp5_96x96_rs0";
#endif
    default: return nullptr;
  }
}

// the inverse of mmh_kernel_name, on the short names the harness, MMULT_KERNEL and the Python API use
int mmh_kernel_id(const char *name) {
  if (!name) return -1;
  const std::string want = std::string("MMult_hip_") + name;
  for (int id = 0; id < 128; ++id) {
    const char *s = mmh_kernel_name(id);
    if (s && (want == s || strcmp(name, s) == 0)) return id;   // (the A/B ids of the tools build carry bare names)
  }
  return -1;
}

int mmh_reserve_stream(mmh_handle_t h, void *stream, int m, int n, int k) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return reserve_stream(h, static_cast<hipStream_t>(stream), m, n, k);
}

int mmh_warm(mmh_handle_t h) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return warm_context(h);
}

int mmh_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
              int ldb, float *dC, int ldc, int accumulate, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate,
                  static_cast<hipStream_t>(stream));
}
int mmh_time_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
                   int ldb, float *dC, int ldc, int warmup, int reps, void *stream,
                   float *ms_per_call) {
  if (!h || reps <= 0 || warmup < 0 || !ms_per_call) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = MMH_OK;
  for (int i = 0; i < warmup && rc == MMH_OK; ++i) rc = sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, s);
  if (rc != MMH_OK) return rc;
  // (every exit below destroys what was created: a sticky error or a failed launch inside the loop must not leak events)
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipError_t e = hipEventCreate(&t0);
  if (e == hipSuccess) e = hipEventCreate(&t1);
  if (e == hipSuccess) e = hipEventRecord(t0, s);
  for (int i = 0; i < reps && rc == MMH_OK && e == hipSuccess; ++i)
    rc = sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, s);
  float ms = 0.f;
  if (rc == MMH_OK && e == hipSuccess) e = hipEventRecord(t1, s);
  if (rc == MMH_OK && e == hipSuccess) e = hipEventSynchronize(t1);
  if (rc == MMH_OK && e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
  if (t0) (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  if (rc != MMH_OK) return rc;
  if (e != hipSuccess) return hip_fail(e, "mmh_time_sgemm");
  *ms_per_call = ms / reps;
  return check_sticky(h);
}

// per-launch durations of `count` back-to-back calls (one event pair each): the clock-ramp trace
int mmh_trace_sgemm(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                    float *dC, int ldc, int count, void *stream, float *ms_each) {
  if (!h || count <= 0 || count > 4096 || !ms_each) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(count + 1, nullptr);
  int rc = MMH_OK;
  hipError_t e = hipSuccess;
  for (auto &x : ev)
    if (e == hipSuccess) e = hipEventCreate(&x);
  if (e == hipSuccess) e = hipEventRecord(ev[0], s);
  for (int i = 0; i < count && rc == MMH_OK && e == hipSuccess; ++i) {
    rc = sgemm_on(h, h->kernel, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, s);
    if (rc == MMH_OK) e = hipEventRecord(ev[i + 1], s);
  }
  if (rc == MMH_OK && e == hipSuccess) e = hipEventSynchronize(ev[count]);
  for (int i = 0; i < count && rc == MMH_OK && e == hipSuccess; ++i) e = hipEventElapsedTime(&ms_each[i], ev[i], ev[i + 1]);
  for (auto &x : ev)   // every event that was created, whatever failed in between
    if (x) (void)hipEventDestroy(x);
  if (rc == MMH_OK && e != hipSuccess) rc = hip_fail(e, "mmh_trace_sgemm");
  return rc;
}

int mmh_streamk_plan(long tiles, int nk, int grid, int *order, int *place) {
  if (!order || !place) return MMH_ERR_INVALID_ARG;
  return build_sk_tables(tiles, nk, grid, order, place) ? MMH_OK : MMH_ERR_INVALID_ARG;
}

int mmh_auto_plan(int m, int n, int k, int lda, int ldb, int ldc, int base_align, int cu_count, int *kernel, long *tiles,
                  int *streamk_grid) {
  return mmh::auto_plan(m, n, k, lda, ldb, ldc, base_align, cu_count, kernel, tiles, streamk_grid);
}

}  // extern "C"
