// utils.cpp -- see utils.h.  Compiled with -ffp-contract=off (makefile).
#include "utils.h"

#include <sys/time.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
