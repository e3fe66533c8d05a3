// utils.cpp -- see utils.h.  Compiled with -ffp-contract=off (makefile).
#include "utils.h"

#include <dlfcn.h>
#include <glob.h>
#include <sys/time.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace {

// rows [r0, r1) of C += A*B; loop order i, p, j keeps each C(i,j)'s products
// in ascending p (bit-identical to the i, j, p loop) but streams B by rows.
void ref_rows(int r0, int r1, int n, int k, const float *a, int lda, const float *b, int ldb,
              float *c, int ldc) {
  for (int i = r0; i < r1; ++i) {
    float *ci = c + (size_t)i * ldc;
    for (int p = 0; p < k; ++p) {
      const float aip = a[(size_t)i * lda + p];
      const float *bp = b + (size_t)p * ldb;
      for (int j = 0; j < n; ++j) ci[j] = ci[j] + aip * bp[j];
    }
  }
}

}  // namespace

int REF_MMult_threads() {
  if (const char *e = std::getenv("REF_THREADS")) return std::max(1, std::atoi(e));
  const unsigned hw = std::thread::hardware_concurrency();
  return hw ? (int)hw : 1;
}

void REF_MMult(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c, int ldc) {
  const int nt = std::min(REF_MMult_threads(), std::max(1, m));
  if (nt == 1) return ref_rows(0, m, n, k, a, lda, b, ldb, c, ldc);
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) {
    const int r0 = (int)((long long)m * t / nt), r1 = (int)((long long)m * (t + 1) / nt);
    pool.emplace_back(ref_rows, r0, r1, n, k, a, lda, b, ldb, c, ldc);
  }
  for (auto &th : pool) th.join();
}

void REF_MMult_serial(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c,
                      int ldc) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      float *cij = c + (size_t)i * ldc + j;
      for (int p = 0; p < k; ++p) *cij = *cij + a[(size_t)i * lda + p] * b[(size_t)p * ldb + j];
    }
}

namespace {

// cblas_sgemm with 32-bit and with 64-bit (ILP64 builds: numpy's bundled OpenBLAS) integer arguments
using sgemm32_t = void (*)(int, int, int, int, int, int, float, const float *, int, const float *, int, float,
                           float *, int);
using sgemm64_t = void (*)(int, int, int, long long, long long, long long, float, const float *, long long,
                           const float *, long long, float, float *, long long);
struct HostBlas {
  void *lib = nullptr;
  sgemm32_t f32 = nullptr;
  sgemm64_t f64 = nullptr;
  std::string path;
};

bool try_blas(HostBlas &hb, const char *path) {
  void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) return false;
  for (const char *sym : {"cblas_sgemm", "scipy_cblas_sgemm"})
    if (void *p = dlsym(lib, sym)) {
      hb.lib = lib, hb.f32 = reinterpret_cast<sgemm32_t>(p), hb.path = path;
      return true;
    }
  for (const char *sym : {"cblas_sgemm64_", "scipy_cblas_sgemm64_"})
    if (void *p = dlsym(lib, sym)) {
      hb.lib = lib, hb.f64 = reinterpret_cast<sgemm64_t>(p), hb.path = path;
      return true;
    }
  dlclose(lib);
  return false;
}

HostBlas &host_blas() {
  static HostBlas hb = [] {
    HostBlas h;
    if (const char *e = std::getenv("MMULT_BLAS_LIB"))
      if (*e && try_blas(h, e)) return h;
    for (const char *name : {"libopenblas.so.0", "libopenblas.so", "libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so",
                             "libblis.so"})
      if (try_blas(h, name)) return h;
    for (const char *pat : {"/usr/local/lib/python3*/dist-packages/scipy.libs/libscipy_openblas*.so",
                            "/usr/local/lib/python3*/dist-packages/numpy.libs/libscipy_openblas*.so",
                            "/usr/lib/python3*/site-packages/scipy.libs/libscipy_openblas*.so",
                            "/usr/lib/python3/dist-packages/numpy.libs/libscipy_openblas*.so",
                            "/opt/conda/lib/libmkl_rt.so*", "/opt/conda/lib/libopenblas*.so*"}) {
      glob_t g{};
      if (glob(pat, 0, nullptr, &g) == 0)
        for (size_t i = 0; i < g.gl_pathc && !h.lib; ++i) try_blas(h, g.gl_pathv[i]);
      globfree(&g);
      if (h.lib) return h;
    }
    return h;
  }();
  return hb;
}

}  // namespace

bool REF_MMult_blas(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c, int ldc) {
  HostBlas &hb = host_blas();
  constexpr int row_major = 101, no_trans = 111;   // CblasRowMajor, CblasNoTrans
  if (hb.f32) hb.f32(row_major, no_trans, no_trans, m, n, k, 1.0f, a, lda, b, ldb, 0.0f, c, ldc);
  else if (hb.f64) hb.f64(row_major, no_trans, no_trans, m, n, k, 1.0f, a, lda, b, ldb, 0.0f, c, ldc);
  else return false;
  return true;
}

const char *REF_MMult_blas_library() { return host_blas().path.c_str(); }

float compare_matrices(int m, int n, float *a, int lda, float *b, int ldb) {
  float worst = 0.0f;
  bool reported = false;
  for (int i = 0; i < m; ++i) {
    const float *ra = a + (size_t)i * lda, *rb = b + (size_t)i * ldb;
    for (int j = 0; j < n; ++j) {
      const float d = std::fabs(ra[j] - rb[j]);
      if (d > worst) worst = d;
      if (!reported && worst > 0.5f) {
        std::printf("\n error: i %d  j %d diff %f  got %f  expect %f ", i, j, worst, ra[j], rb[j]);
        reported = true;
      }
    }
  }
  return worst;
}

void random_matrix(int m, int n, float *a, int lda) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) a[(size_t)j * lda + i] = 2.0 * (float)drand48() - 1.0;
}

void pattern_matrix(int m, int n, float *a, int lda, int mod) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j)
      a[(size_t)j * lda + i] = mod == 0 ? 1.0f : (float)((j - i) % mod);
}

void copy_matrix(int m, int n, float *a, int lda, float *b, int ldb) {
  for (int i = 0; i < m; ++i) std::copy(a + (size_t)i * lda, a + (size_t)i * lda + n, b + (size_t)i * ldb);
}

void print_matrix(int m, int n, float *a, int lda) {
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < n; ++j) std::printf("%.1f\t", a[(size_t)i * lda + j]);
    std::printf("\n");
  }
  std::printf("\n");
}

double dclock() {
  static double t0 = 0.0;
  timeval tv;
  gettimeofday(&tv, nullptr);
  if (t0 == 0.0) t0 = (double)tv.tv_sec;
  return ((double)tv.tv_sec - t0) + tv.tv_usec * 1.0e-6;
}
