// quant_s8.hpp -- the callers either side of the int8 GEMM:
// symmetric per-tensor quantisation fp32 -> int8 and dequantisation int32 -> fp32.
//
// Contract (the reference only has prose for this, README.md:71-85 -- chgemm
// "symmetric quantisation", inputs in [-127,127]; parity unpinned):
//   scale = 127 / max|x|   over the FINITE elements (1 if that maximum is 0; clamped to FLT_MAX when
//                           the maximum is so small -- below ~3.7e-37 -- that the quotient overflows)
//   q     = clamp(rint(x * scale), -127, 127)          round-half-even, never -128;
//                           NaN -> 0, +-inf -> +-127 (non-finite inputs never poison the scale)
//   C_f32 = (float)acc_i32 * (1 / (scale_a * scale_b))
// The kernels are HBM-bound streaming passes: rows to workgroups, 16-byte loads (eight in flight per
// thread in the abs-max pass), one atomic per workgroup for the abs-max, spread over AMAX_WORDS words
// per tensor so that thousands of workgroups do not serialise on one; A and B of a quantised GEMM
// share one launch each.  The dequantisation
// normally runs inside the int8 GEMM's epilogue (igemm_s8.hpp, `deq`); dequantize_kernel is the
// stand-alone form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmh {

typedef float qf32x4 __attribute__((ext_vector_type(4)));
typedef int qi32x4 __attribute__((ext_vector_type(4)));

constexpr int AMAX_WORDS = 64;   // abs-max accumulator words per tensor (reduced by the quantise pass)

// |x| for the abs-max: non-finite elements do not take part (NaN never wins an fmaxf; inf is dropped)
__device__ __forceinline__ float finite_abs(float x) {
  const float a = fabsf(x);
  return a <= 3.402823466e+38f ? a : 0.0f;
}
__device__ __forceinline__ float abs4(qf32x4 v) {
  return fmaxf(fmaxf(finite_abs(v[0]), finite_abs(v[1])), fmaxf(finite_abs(v[2]), finite_abs(v[3])));
}

// One launch covers up to two tensors (blockIdx.y): A and B of a quantised GEMM.
struct QuantTensor {
  const float *x;   // rows x cols window of a row-major matrix with leading dimension ld
  int rows, cols, ld;
  int8_t *q;        // quantised image (quantize_kernel only), leading dimension ldq
  int ldq;
};

// Rows are dealt out to workgroups, columns to threads: no division per element, and when the rows
// are 16-byte aligned (`vec`: x 16-byte aligned, ld % 4 == 0) four columns per float4 load.
__global__ void __launch_bounds__(256) absmax_kernel(QuantTensor ta, QuantTensor tb, int vec_a, int vec_b,
                                                     unsigned *__restrict__ out_bits) {
  const QuantTensor t = blockIdx.y ? tb : ta;
  const bool vec = blockIdx.y ? vec_b : vec_a;
  float best = 0.0f;
  const int c4 = vec ? t.cols / 4 : 0;
  for (int r = blockIdx.x; r < t.rows; r += gridDim.x) {
    const float *row = t.x + (size_t)r * t.ld;
    int c = threadIdx.x;
    for (; c + 1792 < c4; c += 2048) {   // eight independent 16-byte loads in flight per thread
      qf32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const qf32x4 *>(row + 4 * (c + 256 * j));
#pragma unroll
      for (int j = 0; j < 8; ++j) best = fmaxf(best, abs4(v[j]));
    }
    for (; c + 768 < c4; c += 1024) {    // four
      qf32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const qf32x4 *>(row + 4 * (c + 256 * j));
#pragma unroll
      for (int j = 0; j < 4; ++j) best = fmaxf(best, abs4(v[j]));
    }
    for (; c < c4; c += 256) best = fmaxf(best, abs4(*reinterpret_cast<const qf32x4 *>(row + 4 * c)));
    for (c = 4 * c4 + threadIdx.x; c < t.cols; c += 256) best = fmaxf(best, finite_abs(row[c]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_down(best, off, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    // non-negative floats order like their bits; AMAX_WORDS words per tensor share the arrivals
    atomicMax(out_bits + blockIdx.y * AMAX_WORDS + (blockIdx.x % AMAX_WORDS), __float_as_uint(best));
  }
}

// scale from the AMAX_WORDS partial maxima of tensor `y` (every wave computes it for itself)
__device__ __forceinline__ float quant_scale(const unsigned *__restrict__ amax_bits, int y) {
  float amax = __uint_as_float(amax_bits[y * AMAX_WORDS + (threadIdx.x & (AMAX_WORDS - 1))]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  if (!(amax > 0.0f)) return 1.0f;
  const float s = 127.0f / amax;
  return s <= 3.402823466e+38f ? s : 3.402823466e+38f;
}

__device__ __forceinline__ int quantize_one(float x, float scale) {
  const float y = x == x ? x * scale : 0.0f;   // NaN -> 0 (0 * inf cannot occur: scale is finite)
  return (int)fminf(fmaxf(rintf(y), -127.0f), 127.0f);
}

// `vec`: additionally q 4-byte aligned and ldq % 4 == 0 -> one dword of four int8 per float4.
__global__ void __launch_bounds__(256) quantize_kernel(QuantTensor ta, QuantTensor tb, int vec_a, int vec_b,
                                                       const unsigned *__restrict__ amax_bits,
                                                       float *__restrict__ scale_out) {
  const QuantTensor t = blockIdx.y ? tb : ta;
  const bool vec = blockIdx.y ? vec_b : vec_a;
  const float scale = quant_scale(amax_bits, blockIdx.y);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[blockIdx.y] = scale;
  const int c4 = vec ? t.cols / 4 : 0;
  for (int r = blockIdx.x; r < t.rows; r += gridDim.x) {
    const float *row = t.x + (size_t)r * t.ld;
    int8_t *qrow = t.q + (size_t)r * t.ldq;
    for (int c = threadIdx.x; c < c4; c += 256) {
      const qf32x4 v = *reinterpret_cast<const qf32x4 *>(row + 4 * c);
      const unsigned w = (unsigned)(quantize_one(v[0], scale) & 255) | ((unsigned)(quantize_one(v[1], scale) & 255) << 8) |
                         ((unsigned)(quantize_one(v[2], scale) & 255) << 16) |
                         ((unsigned)(quantize_one(v[3], scale) & 255) << 24);
      *reinterpret_cast<unsigned *>(qrow + 4 * c) = w;
    }
    for (int c = 4 * c4 + threadIdx.x; c < t.cols; c += 256) qrow[c] = (int8_t)quantize_one(row[c], scale);
  }
}

inline bool quant_vec_ok(const QuantTensor &t, bool with_q) {
  return ((reinterpret_cast<uintptr_t>(t.x) & 15) == 0) && (t.ld % 4 == 0) &&
         (!with_q || (((reinterpret_cast<uintptr_t>(t.q) & 3) == 0) && (t.ldq % 4 == 0)));
}
// workgroups per tensor: enough to fill the chip several times over, never more than rows.  The
// abs-max pass ends in one atomicMax per workgroup (8192 of them on ONE word serialised into ~80 us
// at N = 4096; they are spread over AMAX_WORDS words and the pass runs with `cap` = 2048).
inline unsigned quant_rows_grid(int rows_a, int rows_b, int cap = 4096) {
  const int r = rows_a > rows_b ? rows_a : rows_b;
  return (unsigned)(r < 1 ? 1 : (r < cap ? r : cap));
}

__global__ void __launch_bounds__(256) dequantize_kernel(const int32_t *__restrict__ acc, int rows,
                                                         int cols, int ldacc,
                                                         const float *__restrict__ scale_a,
                                                         const float *__restrict__ scale_b,
                                                         float *__restrict__ c, int ldc) {
  const float inv = 1.0f / (*scale_a * *scale_b);
  for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int col = threadIdx.x; col < cols; col += 256)
      c[(size_t)r * ldc + col] = (float)acc[(size_t)r * ldacc + col] * inv;
}

}  // namespace mmh
