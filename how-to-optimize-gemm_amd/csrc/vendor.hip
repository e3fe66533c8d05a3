// vendor.hip -- the vendor-library bridges, all loaded with dlopen on first use so that the core library has no
// link-time dependency on any of them:
//   rocBLAS    the reference's first comparator, cuda/MMult_cuBLAS_1.cpp:11-19 (cublasSgemm)   -> mmh_sgemm_rocblas
//   hipBLASLt  its second one, cuda/MMult_cuBLAS_2.cpp:11-26 (cublasGemmEx, fp32 compute)        -> mmh_sgemm_hipblaslt
//   RCCL       the single-process row-panel shard's one broadcast (shard.hip)
// A column-major library computes C^T = B^T * A^T, which is row-major C = A * B (the swapped-operand trick
// of cuda/MMult_cuBLAS_1.cpp:17-18).  Part of libmmult_hip.so (see internal.hpp).
#include <dlfcn.h>
// hipBLASLt's header supplies types and enumerators only (every entry point is looked up at run time); a ROCm install
// without the hipBLASLt development files still builds the library -- mmh_sgemm_hipblaslt then returns
// MMH_ERR_UNSUPPORTED, as it does when the shared object is missing at run time.
#if defined(__has_include)
#if __has_include(<hipblaslt/hipblaslt.h>)
#define MMH_HAVE_HIPBLASLT_H 1
#include <hipblaslt/hipblaslt.h>
#endif
#endif

#include <new>

#include "internal.hpp"

namespace mmh {

// ---------------------------------------------------------------- rocBLAS --
namespace {
struct RocblasApi {
  void *lib = nullptr;
  int (*create)(void **) = nullptr;
  int (*destroy)(void *) = nullptr;
  int (*set_stream)(void *, hipStream_t) = nullptr;
  int (*sgemm)(void *, int, int, int, int, int, const float *, const float *, int, const float *,
               int, const float *, float *, int) = nullptr;
  bool ok = false;
};

inline RocblasApi &rocblas_api() {
  static RocblasApi api = [] {
    RocblasApi a;
    // (the image's ROCm library by its path FIRST: inside a Python process that has imported torch the bare name resolves to
    // the copy bundled in torch's wheel -- an older ROCm's, see blaslt_api)
    a.lib = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) return a;
    a.create = reinterpret_cast<decltype(a.create)>(dlsym(a.lib, "rocblas_create_handle"));
    a.destroy = reinterpret_cast<decltype(a.destroy)>(dlsym(a.lib, "rocblas_destroy_handle"));
    a.set_stream = reinterpret_cast<decltype(a.set_stream)>(dlsym(a.lib, "rocblas_set_stream"));
    a.sgemm = reinterpret_cast<decltype(a.sgemm)>(dlsym(a.lib, "rocblas_sgemm"));
    a.ok = a.create && a.destroy && a.set_stream && a.sgemm;
    return a;
  }();
  return api;
}
}  // namespace

void rocblas_release(void *&handle) {
  if (handle && rocblas_api().ok) rocblas_api().destroy(handle);
  handle = nullptr;
}

int rocblas_sgemm_rowmajor(void **handle, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                           float *dC, int ldc, void *stream) {
  RocblasApi &api = rocblas_api();
  if (!api.ok) {
    set_last_error("librocblas.so could not be loaded");
    return MMH_ERR_UNSUPPORTED;
  }
  if (!*handle && api.create(handle) != 0) {
    set_last_error("rocblas_create_handle failed");
    return MMH_ERR_UNSUPPORTED;
  }
  api.set_stream(*handle, static_cast<hipStream_t>(stream));
  const float one = 1.0f, zero = 0.0f;
  constexpr int op_none = 111;  // rocblas_operation_none
  const int st = api.sgemm(*handle, op_none, op_none, n, m, k, &one, dB, ldb, dA, lda, &zero, dC, ldc);
  if (st != 0) {
    set_last_error("rocblas_sgemm returned status " + std::to_string(st));
    return MMH_ERR_HIP;
  }
  return MMH_OK;
}

// -------------------------------------------------------------- hipBLASLt --
#ifdef MMH_HAVE_HIPBLASLT_H
namespace {
struct BlasLtApi {
  void *lib = nullptr;
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtDestroy) destroy = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescDestroy) desc_destroy = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatrixLayoutDestroy) layout_destroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
  bool ok = false;
};

BlasLtApi &blaslt_api() {
  static BlasLtApi api = [] {
    BlasLtApi a;
    // The image's ROCm library by its path FIRST.  `dlopen("libhipblaslt.so")` inside a Python process that has imported torch
    // returns the copy bundled in torch's wheel (ROCm 7.0 beside the image's 7.2): 15-20 % slower on its stream-K sizes
    // (N = 2560: 122.6 against 147.9 TFLOP/s) -- what every in-process comparison of rounds 1-5 measured (round 6 notes).
    // MMH_VENDOR_BARE_NAMES=1: the old order (whatever the process already holds).
    const char *bare = getenv("MMH_VENDOR_BARE_NAMES");
    if (!(bare && *bare == '1')) a.lib = dlopen("/opt/rocm/lib/libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) return a;
#define MMH_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name))
    MMH_SYM(create, "hipblasLtCreate");
    MMH_SYM(destroy, "hipblasLtDestroy");
    MMH_SYM(desc_create, "hipblasLtMatmulDescCreate");
    MMH_SYM(desc_destroy, "hipblasLtMatmulDescDestroy");
    MMH_SYM(layout_create, "hipblasLtMatrixLayoutCreate");
    MMH_SYM(layout_destroy, "hipblasLtMatrixLayoutDestroy");
    MMH_SYM(pref_create, "hipblasLtMatmulPreferenceCreate");
    MMH_SYM(pref_destroy, "hipblasLtMatmulPreferenceDestroy");
    MMH_SYM(pref_set, "hipblasLtMatmulPreferenceSetAttribute");
    MMH_SYM(heuristic, "hipblasLtMatmulAlgoGetHeuristic");
    MMH_SYM(matmul, "hipblasLtMatmul");
#undef MMH_SYM
    a.ok = a.create && a.destroy && a.desc_create && a.desc_destroy && a.layout_create && a.layout_destroy &&
           a.pref_create && a.pref_destroy && a.pref_set && a.heuristic && a.matmul;
    return a;
  }();
  return api;
}

// per handle: the library handle, a workspace, and the plans (descriptor, layouts, chosen algorithm) of the
// shapes seen so far -- the heuristic query costs far more than a small GEMM, a sweep asks it once per shape
struct BlasLtPlan {
  int m, n, k, lda, ldb, ldc;
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
};
struct BlasLtState {
  hipblasLtHandle_t handle = nullptr;
  DevBuf workspace;
  std::vector<BlasLtPlan> plans;
};
constexpr size_t kBlasLtWorkspace = 64u << 20;

void destroy_plan(BlasLtApi &api, BlasLtPlan &p) {
  if (p.la) api.layout_destroy(p.la);
  if (p.lb) api.layout_destroy(p.lb);
  if (p.lc) api.layout_destroy(p.lc);
  if (p.desc) api.desc_destroy(p.desc);
  p.la = p.lb = p.lc = nullptr;
  p.desc = nullptr;
}
}  // namespace

void hipblaslt_release(void *&state) {
  auto *st = static_cast<BlasLtState *>(state);
  if (!st) return;
  BlasLtApi &api = blaslt_api();
  if (api.ok) {
    for (auto &p : st->plans) destroy_plan(api, p);
    if (st->handle) api.destroy(st->handle);
  }
  st->workspace.release();
  delete st;
  state = nullptr;
}

int hipblaslt_sgemm_rowmajor(void **state, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                             float *dC, int ldc, void *stream) {
  BlasLtApi &api = blaslt_api();
  if (!api.ok) {
    set_last_error("libhipblaslt.so could not be loaded (or lacks an entry point the comparator needs)");
    return MMH_ERR_UNSUPPORTED;
  }
  auto *st = static_cast<BlasLtState *>(*state);
  if (!st) {
    st = new (std::nothrow) BlasLtState;
    if (!st) return MMH_ERR_ALLOC;
    if (api.create(&st->handle) != HIPBLAS_STATUS_SUCCESS) {
      delete st;
      set_last_error("hipblasLtCreate failed");
      return MMH_ERR_UNSUPPORTED;
    }
    if (st->workspace.reserve(kBlasLtWorkspace) != MMH_OK) {
      api.destroy(st->handle);
      delete st;
      return MMH_ERR_ALLOC;
    }
    *state = st;
  }
  BlasLtPlan *plan = nullptr;
  for (auto &p : st->plans)
    if (p.m == m && p.n == n && p.k == k && p.lda == lda && p.ldb == ldb && p.ldc == ldc) plan = &p;
  if (!plan) {
    BlasLtPlan p;
    p.m = m; p.n = n; p.k = k; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    // column-major view: "A" := our B (n x k, ld = ldb), "B" := our A (k x m, ld = lda), C^T (n x m, ld = ldc)
    bool good = api.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS &&
                api.layout_create(&p.la, HIP_R_32F, (uint64_t)n, (uint64_t)k, ldb) == HIPBLAS_STATUS_SUCCESS &&
                api.layout_create(&p.lb, HIP_R_32F, (uint64_t)k, (uint64_t)m, lda) == HIPBLAS_STATUS_SUCCESS &&
                api.layout_create(&p.lc, HIP_R_32F, (uint64_t)n, (uint64_t)m, ldc) == HIPBLAS_STATUS_SUCCESS;
    hipblasLtMatmulPreference_t pref = nullptr;
    good = good && api.pref_create(&pref) == HIPBLAS_STATUS_SUCCESS;
    if (good) {
      uint64_t ws = kBlasLtWorkspace;
      good = api.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws) == HIPBLAS_STATUS_SUCCESS;
    }
    hipblasLtMatmulHeuristicResult_t res[1];
    int found = 0;
    if (good)
      good = api.heuristic(st->handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS &&
             found > 0 && res[0].state == HIPBLAS_STATUS_SUCCESS;
    if (pref) api.pref_destroy(pref);
    if (!good) {
      destroy_plan(api, p);
      set_last_error("hipBLASLt found no algorithm for this fp32 shape");
      return MMH_ERR_UNSUPPORTED;
    }
    p.algo = res[0].algo;
    p.ws = res[0].workspaceSize;
    if (st->plans.size() >= 128) {   // bounded: drop the oldest plan
      destroy_plan(api, st->plans.front());
      st->plans.erase(st->plans.begin());
    }
    st->plans.push_back(p);
    plan = &st->plans.back();
  }
  const float one = 1.0f, zero = 0.0f;
  const hipblasStatus_t rc = api.matmul(st->handle, plan->desc, &one, dB, plan->la, dA, plan->lb, &zero, dC, plan->lc, dC,
                                        plan->lc, &plan->algo, st->workspace.p, st->workspace.bytes,
                                        static_cast<hipStream_t>(stream));
  if (rc != HIPBLAS_STATUS_SUCCESS) {
    set_last_error("hipblasLtMatmul returned status " + std::to_string((int)rc));
    return MMH_ERR_HIP;
  }
  return MMH_OK;
}
#else   // built without <hipblaslt/hipblaslt.h>: the comparator is absent, nothing else is
void hipblaslt_release(void *&state) { state = nullptr; }
int hipblaslt_sgemm_rowmajor(void **, int, int, int, const float *, int, const float *, int, float *, int, void *) {
  set_last_error("this build of libmmult_hip.so was compiled without the hipBLASLt headers: no hipBLASLt comparator");
  return MMH_ERR_UNSUPPORTED;
}
#endif

// ------------------------------------------------------------------- RCCL --
RcclApi &rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    a.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) return a;
    a.get_version = reinterpret_cast<decltype(a.get_version)>(dlsym(a.lib, "ncclGetVersion"));
    a.comm_init_all = reinterpret_cast<decltype(a.comm_init_all)>(dlsym(a.lib, "ncclCommInitAll"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(a.lib, "ncclCommDestroy"));
    a.comm_count = reinterpret_cast<decltype(a.comm_count)>(dlsym(a.lib, "ncclCommCount"));
    a.group_start = reinterpret_cast<decltype(a.group_start)>(dlsym(a.lib, "ncclGroupStart"));
    a.group_end = reinterpret_cast<decltype(a.group_end)>(dlsym(a.lib, "ncclGroupEnd"));
    a.broadcast = reinterpret_cast<decltype(a.broadcast)>(dlsym(a.lib, "ncclBroadcast"));
    a.ok = a.get_version && a.comm_init_all && a.comm_destroy && a.comm_count && a.group_start && a.group_end &&
           a.broadcast;
    return a;
  }();
  return api;
}

}  // namespace mmh

using namespace mmh;

extern "C" {

int mmh_sgemm_rocblas(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb, float *dC,
                      int ldc, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  ENTER(h);
  if (m == 0 || n == 0 || k == 0)
    return sgemm_on(h, MMH_KERNEL_MFMA, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, static_cast<hipStream_t>(stream));
  return rocblas_sgemm_rowmajor(&h->rocblas, m, n, k, dA, lda, dB, ldb, dC, ldc, stream);
}

int mmh_sgemm_hipblaslt(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                        float *dC, int ldc, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  ENTER(h);
  if (m == 0 || n == 0 || k == 0)
    return sgemm_on(h, MMH_KERNEL_MFMA, m, n, k, dA, lda, dB, ldb, dC, ldc, 0, static_cast<hipStream_t>(stream));
  return hipblaslt_sgemm_rowmajor(&h->blaslt, m, n, k, dA, lda, dB, ldb, dC, ldc, stream);
}

// mmh_time_sgemm for the comparators: the same event pair around `reps` back-to-back calls issued from C
int mmh_time_comparator(mmh_handle_t h, int which, int m, int n, int k, const float *dA, int lda, const float *dB, int ldb,
                        float *dC, int ldc, int warmup, int reps, void *stream, float *ms_per_call) {
  if (!h || reps <= 0 || warmup < 0 || !ms_per_call || (which != MMH_COMPARATOR_ROCBLAS && which != MMH_COMPARATOR_HIPBLASLT))
    return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK || m == 0 || n == 0 || k == 0) return rc != MMH_OK ? rc : MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto call = [&] {
    return which == MMH_COMPARATOR_ROCBLAS ? rocblas_sgemm_rowmajor(&h->rocblas, m, n, k, dA, lda, dB, ldb, dC, ldc, stream)
                                           : hipblaslt_sgemm_rowmajor(&h->blaslt, m, n, k, dA, lda, dB, ldb, dC, ldc, stream);
  };
  for (int i = 0; i < warmup; ++i)
    if ((rc = call()) != MMH_OK) return rc;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  HIP_TRY(hipEventCreate(&t0));
  hipError_t e = hipEventCreate(&t1);
  if (e == hipSuccess) e = hipEventRecord(t0, s);
  for (int i = 0; i < reps && rc == MMH_OK && e == hipSuccess; ++i) rc = call();
  if (e == hipSuccess) e = hipEventRecord(t1, s);
  if (e == hipSuccess) e = hipEventSynchronize(t1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, t0, t1);
  (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  if (rc != MMH_OK) return rc;
  if (e != hipSuccess) return hip_fail(e, "mmh_time_comparator");
  *ms_per_call = ms / reps;
  return MMH_OK;
}

int mmh_rccl_version(int *version) {
  if (!version) return MMH_ERR_INVALID_ARG;
  *version = 0;
  RcclApi &api = rccl_api();
  if (!api.ok) {
    set_last_error("librccl.so could not be loaded (or lacks an entry point the shard needs)");
    return MMH_ERR_UNSUPPORTED;
  }
  if (api.get_version(version) != 0) return MMH_ERR_COMM;
  return MMH_OK;
}

}  // extern "C"
