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
// The kernels are HBM-bound streaming passes: QG contiguous 16 KiB row chunks per workgroup, 16-byte loads, one
// atomic per workgroup for the abs-max, spread over AMAX_WORDS words per tensor so that thousands of
// workgroups do not serialise on one; A and B of a quantised GEMM share one launch each.  The dequantisation
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

// Access pattern (both passes): with 16-byte aligned rows (`vec`: x 16-byte aligned, ld % 4 == 0) every
// workgroup owns QG consecutive 16 KiB chunks of the tensor (a chunk = 256 threads x QU float4 of one
// row) and walks them with the loads of chunk g+1 issued before chunk g is consumed; the grid is as
// large as the tensor needs and nobody loops further.  Contiguous chunks per workgroup are what streams
// fastest on this part (tools/probes/hbm_patterns.hip: 6.0-6.5 TB/s against 5.0 for a persistent
// grid-stride loop); QG > 1 because a workgroup that lives for ONE 16 KiB chunk spends as long in its
// reduction / scale prologue as in its loads (measured: 35 us for the 134 MB abs-max pass of a 4096^3
// quantised GEMM with QG = 1, 26.8 us for the round-1 row loop).  Unaligned tensors take the scalar row loop.
constexpr int QU = 4;                    // float4 per thread and chunk
constexpr int QCHUNK = 256 * QU;         // float4 per chunk
constexpr int QG = 4;                    // chunks per workgroup

__host__ __device__ inline int quant_chunks_per_row(int cols) { return (cols / 4 + QCHUNK - 1) / QCHUNK; }

// chunk `id` (0 .. rows * cpr - 1) of tensor t -> its row pointer offset and first float4 column of this thread
struct QuantChunk {
  int row, c0;
  bool live;
};
__device__ __forceinline__ QuantChunk quant_chunk(const QuantTensor &t, int cpr, long id) {
  QuantChunk q;
  q.row = (int)(id / cpr);
  q.c0 = (int)(id - (long)q.row * cpr) * QCHUNK + threadIdx.x;
  q.live = q.row < t.rows;
  return q;
}
__device__ __forceinline__ void quant_load(const QuantTensor &t, const QuantChunk &q, int c4, qf32x4 (&v)[QU]) {
  const float *row = t.x + (size_t)q.row * t.ld;
#pragma unroll
  for (int j = 0; j < QU; ++j)
    v[j] = (q.live && q.c0 + 256 * j < c4) ? *reinterpret_cast<const qf32x4 *>(row + 4 * (q.c0 + 256 * j))
                                            : qf32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ void __launch_bounds__(256) absmax_kernel(QuantTensor ta, QuantTensor tb, int vec_a, int vec_b,
                                                     unsigned *__restrict__ out_bits) {
  const QuantTensor t = blockIdx.y ? tb : ta;
  const bool vec = blockIdx.y ? vec_b : vec_a;
  float best = 0.0f;
  if (vec) {
    const int c4 = t.cols / 4, cpr = quant_chunks_per_row(t.cols);
    const long first = (long)blockIdx.x * QG;
    if (first >= (long)t.rows * cpr) return;
    qf32x4 cur[QU], nxt[QU];
    quant_load(t, quant_chunk(t, cpr, first), c4, cur);
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const QuantChunk q = quant_chunk(t, cpr, first + g);
      if (g + 1 < QG) quant_load(t, quant_chunk(t, cpr, first + g + 1), c4, nxt);
#pragma unroll
      for (int j = 0; j < QU; ++j) best = fmaxf(best, abs4(cur[j]));
      if (q.live && q.c0 - (int)threadIdx.x == 0)       // first chunk of a row: the (< 4) columns past the float4s
        for (int c = 4 * c4 + threadIdx.x; c < t.cols; c += 256) best = fmaxf(best, finite_abs(t.x[(size_t)q.row * t.ld + c]));
#pragma unroll
      for (int j = 0; j < QU; ++j) cur[j] = nxt[j];
    }
  } else {
    for (int r = blockIdx.x; r < t.rows; r += gridDim.x) {
      const float *row = t.x + (size_t)r * t.ld;
      for (int c = threadIdx.x; c < t.cols; c += 256) best = fmaxf(best, finite_abs(row[c]));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_down(best, off, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    // non-negative floats order like their bits; AMAX_WORDS words per tensor share the arrivals, and a
    // workgroup whose maximum is 0 has nothing to say
    if (best > 0.0f) atomicMax(out_bits + blockIdx.y * AMAX_WORDS + (blockIdx.x % AMAX_WORDS), __float_as_uint(best));
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
__device__ __forceinline__ unsigned quantize4(qf32x4 v, float scale) {
  return (unsigned)(quantize_one(v[0], scale) & 255) | ((unsigned)(quantize_one(v[1], scale) & 255) << 8) |
         ((unsigned)(quantize_one(v[2], scale) & 255) << 16) | ((unsigned)(quantize_one(v[3], scale) & 255) << 24);
}

__global__ void __launch_bounds__(256) quantize_kernel(QuantTensor ta, QuantTensor tb, int vec_a, int vec_b,
                                                       const unsigned *__restrict__ amax_bits,
                                                       float *__restrict__ scale_out) {
  const QuantTensor t = blockIdx.y ? tb : ta;
  const bool vec = blockIdx.y ? vec_b : vec_a;
  if (vec) {
    const int c4 = t.cols / 4, cpr = quant_chunks_per_row(t.cols);
    const long first = (long)blockIdx.x * QG;
    qf32x4 cur[QU], nxt[QU];
    const bool any = first < (long)t.rows * cpr;
    if (any) quant_load(t, quant_chunk(t, cpr, first), c4, cur);   // the first loads go out before the scale is needed
    const float scale = quant_scale(amax_bits, blockIdx.y);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[blockIdx.y] = scale;
    if (!any) return;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const QuantChunk q = quant_chunk(t, cpr, first + g);
      if (g + 1 < QG) quant_load(t, quant_chunk(t, cpr, first + g + 1), c4, nxt);
      if (q.live) {
        int8_t *qrow = t.q + (size_t)q.row * t.ldq;
#pragma unroll
        for (int j = 0; j < QU; ++j)
          if (q.c0 + 256 * j < c4) *reinterpret_cast<unsigned *>(qrow + 4 * (q.c0 + 256 * j)) = quantize4(cur[j], scale);
        if (q.c0 - (int)threadIdx.x == 0)
          for (int c = 4 * c4 + threadIdx.x; c < t.cols; c += 256)
            qrow[c] = (int8_t)quantize_one(t.x[(size_t)q.row * t.ld + c], scale);
      }
#pragma unroll
      for (int j = 0; j < QU; ++j) cur[j] = nxt[j];
    }
  } else {
    const float scale = quant_scale(amax_bits, blockIdx.y);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[blockIdx.y] = scale;
    for (int r = blockIdx.x; r < t.rows; r += gridDim.x) {
      const float *row = t.x + (size_t)r * t.ld;
      int8_t *qrow = t.q + (size_t)r * t.ldq;
      for (int c = threadIdx.x; c < t.cols; c += 256) qrow[c] = (int8_t)quantize_one(row[c], scale);
    }
  }
}

inline bool quant_vec_ok(const QuantTensor &t, bool with_q) {
  return ((reinterpret_cast<uintptr_t>(t.x) & 15) == 0) && (t.ld % 4 == 0) &&
         (!with_q || (((reinterpret_cast<uintptr_t>(t.q) & 3) == 0) && (t.ldq % 4 == 0)));
}
// grid.x of a pass over (up to) two tensors: one workgroup per QG 16 KiB row chunks for aligned tensors, a
// few thousand row-striding workgroups for unaligned ones; the larger of the two counts (surplus
// workgroups of the smaller tensor exit at once)
inline unsigned quant_grid(const QuantTensor &t, bool vec) {
  if (t.rows <= 0 || t.cols <= 0) return 1;
  if (vec) return (unsigned)(((long)t.rows * quant_chunks_per_row(t.cols) + QG - 1) / QG);
  return (unsigned)(t.rows < 4096 ? t.rows : 4096);
}
inline unsigned quant_rows_grid(int rows_a, int rows_b, int cap = 4096) {   // dequantize_kernel's row loop
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
