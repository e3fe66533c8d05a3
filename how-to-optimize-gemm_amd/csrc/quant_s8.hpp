// quant_s8.hpp -- the callers either side of the int8 GEMM:
// symmetric per-tensor quantisation fp32 -> int8 and dequantisation int32 -> fp32.
//
// Contract (the reference only has prose for this, README.md:71-85 -- chgemm
// "symmetric quantisation", inputs in [-127,127]; parity unpinned):
//   scale = 127 / max|x|   (1 if the tensor is all zero)
//   q     = clamp(rint(x * scale), -127, 127)          round-half-even, never -128
//   C_f32 = (float)acc_i32 * (1 / (scale_a * scale_b))
// All three kernels are HBM-bound streaming passes: 16-byte loads, grid-stride,
// one atomic per workgroup for the abs-max.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmh {

typedef float qf32x4 __attribute__((ext_vector_type(4)));
typedef int qi32x4 __attribute__((ext_vector_type(4)));

// rows x cols window of a row-major matrix with leading dimension ld
__global__ void __launch_bounds__(256) absmax_kernel(const float *__restrict__ x, int rows, int cols,
                                                     int ld, unsigned *__restrict__ out_bits) {
  float best = 0.0f;
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols, c = i - r * cols;
    best = fmaxf(best, fabsf(x[r * ld + c]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_down(best, off, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    atomicMax(out_bits, __float_as_uint(best));     // non-negative floats order like their bits
  }
}

__global__ void __launch_bounds__(256) quantize_kernel(const float *__restrict__ x, int rows, int cols,
                                                       int ld, const unsigned *__restrict__ amax_bits,
                                                       int8_t *__restrict__ q, int ldq,
                                                       float *__restrict__ scale_out) {
  const float amax = __uint_as_float(*amax_bits);
  const float scale = amax > 0.0f ? 127.0f / amax : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols, c = i - r * cols;
    float v = rintf(x[r * ld + c] * scale);
    v = fminf(fmaxf(v, -127.0f), 127.0f);
    q[r * ldq + c] = (int8_t)v;
  }
}

__global__ void __launch_bounds__(256) dequantize_kernel(const int32_t *__restrict__ acc, int rows,
                                                         int cols, int ldacc,
                                                         const float *__restrict__ scale_a,
                                                         const float *__restrict__ scale_b,
                                                         float *__restrict__ c, int ldc) {
  const float inv = 1.0f / (*scale_a * *scale_b);
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols, col = i - r * cols;
    c[r * ldc + col] = (float)acc[r * ldacc + col] * inv;
  }
}

inline unsigned quant_grid(size_t total) {
  size_t g = (total + 255) / 256;
  return (unsigned)(g < 2048 ? (g ? g : 1) : 2048);
}

}  // namespace mmh
