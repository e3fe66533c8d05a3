// utils.h -- the harness's L2 utilities: oracle,
// comparison, input generation.  Same names and argument meaning as the
// reference's free functions so that a driver written against the reference
// (cuda/test_MMult.cpp:12-19) compiles against these unchanged.
#pragma once

// C = A*B + C, row-major with leading dimensions; i/j/p summation order, one
// fp32 accumulator per element (armv7/REF_MMult.c:9-22).  Built with
// -ffp-contract=off: multiply and add round separately, like the reference's
// x86 -O2 build.  Rows are split over host threads (the per-element order is
// unchanged, so the result is identical to the serial loop).
void REF_MMult(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c, int ldc);
// Serial, literal triple loop (what is timed as the 1-core CPU baseline).
void REF_MMult_serial(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c, int ldc);
int REF_MMult_threads();
// The cuda directory's oracle (cuda/REF_MMult.cpp:9-13): C = A*B by a host BLAS,
//   cblas_sgemm(CblasRowMajor, CblasNoTrans, CblasNoTrans, m, n, k, 1.0f, a, lda, b, ldb, 0.0f, c, ldc)
// -- a blocked, FMA, multi-threaded summation order, which is what the published diff columns
// (cuda/output_MMult_cuda_12.m:5-29: 7.2e-5 at 1024 ... 3.5e-4 at 4096) are measured against.
// The reference links -lopenblas (cuda/makefile:16); no cblas.h exists in this image, so the symbol
// is looked up at run time (dlopen) with a hand-declared prototype: $MMULT_BLAS_LIB if set, else
// libopenblas / libmkl_rt on the loader path, else the OpenBLAS bundled with numpy / scipy.
// Returns false (and leaves c alone) when no BLAS with a cblas_sgemm could be loaded.
bool REF_MMult_blas(int m, int n, int k, float *a, int lda, float *b, int ldb, float *c, int ldc);
const char *REF_MMult_blas_library();   // path of the library that was loaded ("" if none)

// max |a - b| over an m x n window; prints the first element whose running max
// exceeds 0.5 once (cuda/compare_matrices.cpp:7-30).
float compare_matrices(int m, int n, float *a, int lda, float *b, int ldb);

// drand48 uniform [-1,1), stored a[j*lda+i] (cuda/random_matrix.cpp:3-16).
void random_matrix(int m, int n, float *a, int lda);
// Known-answer patterns: mod 3 / 2 -> (j-i)%mod, 0 -> all ones
// (cuda/random_matrix.cpp:13-14, armv7/random_matrix.c:15, aarch64/random_matrix.cpp:16).
void pattern_matrix(int m, int n, float *a, int lda, int mod);

void copy_matrix(int m, int n, float *a, int lda, float *b, int ldb);
void print_matrix(int m, int n, float *a, int lda);
double dclock();
