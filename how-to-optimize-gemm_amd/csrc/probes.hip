// probes.hip -- peak probes.  The reference measures its ceilings before quoting percentages
// (aarch64/gflops_benchmark/main.c:19-25 -- an FMLA-only loop; vulkan/benchmark/gmem_bandwidth.cpp:8-48 -- a copy
// kernel; vulkan/benchmark/smem_bandwidth.cpp:30-42).  Same idea on gfx950: MFMA-only loops (fp32 and int8, no memory
// traffic), a float4 stream copy / read, LDS fragment reads.  Part of libmmult_hip.so (see internal.hpp).
#include "internal.hpp"
#include "sgemm_tile.hpp"

namespace mmh {
namespace {

#define MMH_HIP_TRY(expr)                                              \
  do {                                                                 \
    hipError_t e_ = (expr);                                            \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);                  \
  } while (0)

// ------------------------------------------------------------- MFMA probe --
__global__ void __launch_bounds__(256) probe_mfma_kernel(float *out, int iters, float seed) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  const float a = seed + threadIdx.x * 1e-9f, b = seed - threadIdx.x * 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];  // keep the chain live
}

int probe_mfma_f32(int cu_count, float *tflops) {
  if (cu_count <= 0) cu_count = 256;
  float *d = nullptr;
  MMH_HIP_TRY(hipMalloc(&d, 64));
  const int iters = 20000, blocks = cu_count * 2;
  hipEvent_t t0, t1;
  MMH_HIP_TRY(hipEventCreate(&t0));
  MMH_HIP_TRY(hipEventCreate(&t1));
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, 0, d, 2000, 0.001f);
  MMH_HIP_TRY(hipEventRecord(t0, 0));
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 0.001f);
  MMH_HIP_TRY(hipEventRecord(t1, 0));
  MMH_HIP_TRY(hipEventSynchronize(t1));
  float ms = 0.f;
  MMH_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
  const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
  *tflops = (float)(flops / (ms * 1e-3) / 1e12);
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  (void)hipFree(d);
  return MMH_OK;
}

// The vector ALU's fp32 FMA rate and nothing else (round 5: the denominator of the K1 / K1W rung): 32 independent
// accumulator pairs per lane, `waves` waves per SIMD, random-ish operands so that the power manager sees real toggling.
// packed: v_pk_fma_f32 (two FMAs per lane and instruction); otherwise v_fma_f32.
template <bool PACKED>
__global__ void __launch_bounds__(1024) probe_valu_kernel(float *out, int iters, float seed) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x2{seed * (float)(i + 1), seed * (float)(threadIdx.x + i)};
  f32x2 a = {1.0f + seed * (float)threadIdx.x, 1.0f - seed * (float)threadIdx.x}, b = {seed, -seed};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if constexpr (PACKED) {
        acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
      } else {
        acc[i][0] = __builtin_fmaf(a[0], b[0], acc[i][0]);
        acc[i][1] = __builtin_fmaf(a[1], b[1], acc[i][1]);
      }
    }
#pragma unroll
    for (int i = 0; i < 32; i += 2) asm volatile("" : "+v"(acc[i]), "+v"(acc[i + 1]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1];
  if (s == 12345.678f) out[0] = s;
}

int probe_valu_f32(int cu_count, int packed, int waves_per_simd, float *tflops) {
  if (cu_count <= 0) cu_count = 256;
  // (every early return releases what was created; a launch that failed must not leave ms at 0 and an
  // infinite rate in the bench line)
  struct Res {
    float *d = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    ~Res() {
      if (t0) (void)hipEventDestroy(t0);
      if (t1) (void)hipEventDestroy(t1);
      if (d) (void)hipFree(d);
    }
  } r;
  MMH_HIP_TRY(hipMalloc(&r.d, 64));
  const int iters = 20000, blocks = cu_count, threads = 256 * waves_per_simd;
  MMH_HIP_TRY(hipEventCreate(&r.t0));
  MMH_HIP_TRY(hipEventCreate(&r.t1));
  auto launch = [&](int n) {
    if (packed) hipLaunchKernelGGL(probe_valu_kernel<true>, dim3(blocks), dim3(threads), 0, 0, r.d, n, 0.001f);
    else hipLaunchKernelGGL(probe_valu_kernel<false>, dim3(blocks), dim3(threads), 0, 0, r.d, n, 0.001f);
    return hipGetLastError();
  };
  for (int w = 0; w < 3; ++w) MMH_HIP_TRY(launch(iters));      // ~10 ms: the power manager's sustained state
  MMH_HIP_TRY(hipEventRecord(r.t0, 0));
  MMH_HIP_TRY(launch(iters));
  MMH_HIP_TRY(hipEventRecord(r.t1, 0));
  MMH_HIP_TRY(hipEventSynchronize(r.t1));
  float ms = 0.f;
  MMH_HIP_TRY(hipEventElapsedTime(&ms, r.t0, r.t1));
  if (!(ms > 0.f)) {
    set_last_error("probe_valu_f32: the timed launch took no measurable time");
    return MMH_ERR_HIP;
  }
  const double flops = (double)blocks * threads * iters * 64.0 * 2.0;
  *tflops = (float)(flops / (ms * 1e-3) / 1e12);
  return MMH_OK;
}

// int8 twin: v_mfma_i32_16x16x64_i8 only (what K3 issues), 8 accumulators per wave
typedef int pi32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) probe_mfma_i8_kernel(int *out, int iters, int seed) {
  // The MFMAs are spelled in inline asm: with the builtin, hipcc 7.2 allocates the eight
  // int accumulators in overlapping AGPR windows and shuffles ~50 registers per iteration,
  // which measures the shuffle, not the matrix pipe (2.4 instead of ~4 POPS).
  pi32x4 acc0 = {seed, seed, seed, seed}, acc1 = acc0, acc2 = acc0, acc3 = acc0, acc4 = acc0, acc5 = acc0,
         acc6 = acc0, acc7 = acc0;
  const pi32x4 a = {seed + (int)threadIdx.x, seed, 1, 2}, b = {seed, 3, (int)threadIdx.x, 4};
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "v_mfma_i32_16x16x64_i8 %0, %8, %9, %0\n\t"
        "v_mfma_i32_16x16x64_i8 %1, %8, %9, %1\n\t"
        "v_mfma_i32_16x16x64_i8 %2, %8, %9, %2\n\t"
        "v_mfma_i32_16x16x64_i8 %3, %8, %9, %3\n\t"
        "v_mfma_i32_16x16x64_i8 %4, %8, %9, %4\n\t"
        "v_mfma_i32_16x16x64_i8 %5, %8, %9, %5\n\t"
        "v_mfma_i32_16x16x64_i8 %6, %8, %9, %6\n\t"
        "v_mfma_i32_16x16x64_i8 %7, %8, %9, %7"
        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
        : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU readers (no auto padding in asm)
  pi32x4 s = acc0 + acc1 + acc2 + acc3 + acc4 + acc5 + acc6 + acc7;
  if (s[0] + s[1] + s[2] + s[3] == 123456789) out[0] = s[0];
}

// Same loop with four different pseudo-random A and B operands per wave (every MFMA sees new
// bits on its inputs, as a GEMM on random data does): the sustained, power-managed rate.
__global__ void __launch_bounds__(256) probe_mfma_i8_random_kernel(int *out, int iters, int seed) {
  pi32x4 acc0 = {seed, seed, seed, seed}, acc1 = acc0, acc2 = acc0, acc3 = acc0, acc4 = acc0, acc5 = acc0,
         acc6 = acc0, acc7 = acc0;
  pi32x4 a[4], b[4];
  unsigned x = (unsigned)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u + 12345u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x = x * 1664525u + 1013904223u;
      a[i][j] = (int)(x ^ (x >> 15));
      x = x * 1664525u + 1013904223u;
      b[i][j] = (int)(x ^ (x >> 13));
    }
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "v_mfma_i32_16x16x64_i8 %0, %8, %12, %0\n\t"
        "v_mfma_i32_16x16x64_i8 %1, %9, %13, %1\n\t"
        "v_mfma_i32_16x16x64_i8 %2, %10, %14, %2\n\t"
        "v_mfma_i32_16x16x64_i8 %3, %11, %15, %3\n\t"
        "v_mfma_i32_16x16x64_i8 %4, %8, %13, %4\n\t"
        "v_mfma_i32_16x16x64_i8 %5, %9, %14, %5\n\t"
        "v_mfma_i32_16x16x64_i8 %6, %10, %15, %6\n\t"
        "v_mfma_i32_16x16x64_i8 %7, %11, %12, %7"
        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  pi32x4 s = acc0 + acc1 + acc2 + acc3 + acc4 + acc5 + acc6 + acc7;
  if (s[0] + s[1] + s[2] + s[3] == 123456789) out[0] = s[0];
}

// random_operands: 0 = the constant-operand loop, 1 = the random-operand loop; the timed launch is
// repeated until `min_ms` have passed and the LAST launch's rate is returned (sustained clock).
int probe_mfma_i8(int cu_count, float *tops, int random_operands = 0,
                         float min_ms = 0.f) {
  if (cu_count <= 0) cu_count = 256;
  int *d = nullptr;
  MMH_HIP_TRY(hipMalloc(&d, 64));
  const int iters = 40000, blocks = cu_count * 2;
  hipEvent_t t0, t1;
  MMH_HIP_TRY(hipEventCreate(&t0));
  MMH_HIP_TRY(hipEventCreate(&t1));
  auto launch = [&](int n) {
    if (random_operands) hipLaunchKernelGGL(probe_mfma_i8_random_kernel, dim3(blocks), dim3(256), 0, 0, d, n, 1);
    else hipLaunchKernelGGL(probe_mfma_i8_kernel, dim3(blocks), dim3(256), 0, 0, d, n, 1);
  };
  launch(4000);
  float ms = 0.f, total = 0.f;
  do {
    MMH_HIP_TRY(hipEventRecord(t0, 0));
    launch(iters);
    MMH_HIP_TRY(hipEventRecord(t1, 0));
    MMH_HIP_TRY(hipEventSynchronize(t1));
    MMH_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    total += ms;
  } while (total < min_ms);
  const double ops = (double)blocks * 4 * iters * 8.0 * (2.0 * 16 * 16 * 64);
  *tops = (float)(ops / (ms * 1e-3) / 1e12);
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  (void)hipFree(d);
  return MMH_OK;
}

// -------------------------------------------------------------- HBM probe --
// Stream copy / read in the access pattern that streams fastest on this part (measured with
// tools/probes/hbm_patterns.hip, profiles/r02_hbm_patterns.txt): every workgroup owns ONE contiguous
// 16 KiB chunk -- thread t moves the float4s base + t + 256 j, j = 0..3, all four loads issued before
// the first store -- and exits; the grid is as large as the buffer (65536 workgroups per GiB).  A
// persistent grid-stride loop over the same buffer (the round-1 form of this probe, and of the
// abs-max / quantise passes) reads 4.3-5.3 TB/s; the chunk form reads 6.3 TB/s for the copy (the
// guide's figure for a float4 copy, MI355X_MICROARCH.md chip-level parameters) and 6.5 TB/s read-only.
constexpr int PROBE_U = 4;   // float4 per thread
__global__ void __launch_bounds__(256) probe_copy_kernel(const f32x4 *__restrict__ src,
                                                         f32x4 *__restrict__ dst, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * PROBE_U + threadIdx.x;
  f32x4 v[PROBE_U];
#pragma unroll
  for (int j = 0; j < PROBE_U; ++j)
    if (base + j * 256 < n) v[j] = __builtin_nontemporal_load(src + base + j * 256);
#pragma unroll
  for (int j = 0; j < PROBE_U; ++j)
    if (base + j * 256 < n) __builtin_nontemporal_store(v[j], dst + base + j * 256);
}

// Read-only twin (what an abs-max style reduction can reach).
__global__ void __launch_bounds__(256) probe_read_kernel(const f32x4 *__restrict__ src, float *__restrict__ out,
                                                         size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * PROBE_U + threadIdx.x;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < PROBE_U; ++j)
    if (base + j * 256 < n) s += src[base + j * 256];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];   // keep the loads live
}

// mode 0: copy (read + write bytes counted); mode 1: read only.
int probe_hbm_copy(size_t bytes, float *gbps, int cu_count = 256, int mode = 0) {
  const size_t n = bytes / sizeof(f32x4);
  f32x4 *src = nullptr, *dst = nullptr;
  MMH_HIP_TRY(hipMalloc(&src, n * sizeof(f32x4)));
  if (hipMalloc(&dst, mode == 0 ? n * sizeof(f32x4) : 64) != hipSuccess) {
    (void)hipFree(src);
    set_last_error("hipMalloc(dst) failed");
    return MMH_ERR_ALLOC;
  }
  MMH_HIP_TRY(hipMemset(src, 1, n * sizeof(f32x4)));
  hipEvent_t t0, t1;
  MMH_HIP_TRY(hipEventCreate(&t0));
  MMH_HIP_TRY(hipEventCreate(&t1));
  (void)cu_count;
  const unsigned blocks = (unsigned)((n + 256 * PROBE_U - 1) / (256 * PROBE_U));
  const int reps = 10;
  auto launch = [&] {
    if (mode == 0) hipLaunchKernelGGL(probe_copy_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, n);
    else hipLaunchKernelGGL(probe_read_kernel, dim3(blocks), dim3(256), 0, 0, src, reinterpret_cast<float *>(dst), n);
  };
  launch();
  launch();
  MMH_HIP_TRY(hipEventRecord(t0, 0));
  for (int r = 0; r < reps; ++r) launch();
  MMH_HIP_TRY(hipEventRecord(t1, 0));
  MMH_HIP_TRY(hipEventSynchronize(t1));
  float ms = 0.f;
  MMH_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
  *gbps = (float)((mode == 0 ? 2.0 : 1.0) * n * sizeof(f32x4) * reps / (ms * 1e-3) / 1e9);
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  (void)hipFree(src);
  (void)hipFree(dst);
  return MMH_OK;
}

// --------------------------------------------------------------- LDS read --
// The third denominator (the idea of vulkan/benchmark/smem_bandwidth.cpp:30-42): fragment reads out
// of LDS and nothing else.  Two waves per SIMD, eight reads in flight per wave, conflict-free
// lane-linear addresses (roof: 256 B/clk/CU for the 8- and 16-byte reads, 128 for ds_read_b32);
// WIDTH = bytes per lane (16: ds_read_b128, 8: ds_read_b64, 4: ds_read_b32,
// -8: ds_read_b64_tr_b8, the transposing read of the in-place int8 kernel).
template <int WIDTH>
__global__ void __launch_bounds__(512) probe_lds_read_kernel(float *__restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lbuf[16384];   // 64 KiB
  for (int i = threadIdx.x; i < 16384; i += 512) lbuf[i] = (uint32_t)i * 2654435761u;
  __syncthreads();
  constexpr int W = WIDTH < 0 ? -WIDTH : WIDTH;
  const uint32_t addr = (uint32_t)(uintptr_t)lbuf + (threadIdx.x & 63) * W + (threadIdx.x >> 6) * 4096;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (WIDTH == 16) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][3];
    } else if constexpr (W == 8) {
      u32x2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (WIDTH > 0) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 512));
        else asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 512));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][1];
    } else {
      uint32_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
  }
  if (acc == 0x12345u) out[0] = 1.0f;   // keep the reads live
}

// bytes per clock per CU are derived by the caller from the device clock; this returns aggregate GB/s
int probe_lds_read(int width, float *gbps, int cu_count) {
  float *out = nullptr;
  MMH_HIP_TRY(hipMalloc(&out, 64));
  hipEvent_t t0, t1;
  MMH_HIP_TRY(hipEventCreate(&t0));
  MMH_HIP_TRY(hipEventCreate(&t1));
  const int iters = 4096, blocks = 2 * cu_count;
  auto launch = [&] {
    switch (width) {
      case 16: hipLaunchKernelGGL(probe_lds_read_kernel<16>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
      case 8: hipLaunchKernelGGL(probe_lds_read_kernel<8>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
      case -8: hipLaunchKernelGGL(probe_lds_read_kernel<-8>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
      default: hipLaunchKernelGGL(probe_lds_read_kernel<4>, dim3(blocks), dim3(512), 0, 0, out, iters); break;
    }
  };
  launch();
  const int reps = 10;
  MMH_HIP_TRY(hipEventRecord(t0, 0));
  for (int r = 0; r < reps; ++r) launch();
  MMH_HIP_TRY(hipEventRecord(t1, 0));
  MMH_HIP_TRY(hipEventSynchronize(t1));
  float ms = 0.f;
  MMH_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
  const int w = width < 0 ? -width : width;
  *gbps = (float)((double)blocks * 512 * 8 * w * iters * reps / (ms * 1e-3) / 1e9);
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  (void)hipFree(out);
  return MMH_OK;
}

}  // namespace
}  // namespace mmh

using namespace mmh;

extern "C" {

int mmh_probe_mfma_f32(mmh_handle_t h, float *tflops) {
  if (!h || !tflops) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_mfma_f32(h->cu_count, tflops);
}

int mmh_probe_valu_f32(mmh_handle_t h, int packed, int waves_per_simd, float *tflops) {
  if (!h || !tflops || waves_per_simd < 1 || waves_per_simd > 4) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_valu_f32(h->cu_count, packed, waves_per_simd, tflops);
}

int mmh_probe_mfma_i8(mmh_handle_t h, float *tops) {
  if (!h || !tops) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_mfma_i8(h->cu_count, tops);
}

int mmh_probe_mfma_i8_sustained(mmh_handle_t h, int random_operands, float min_ms, float *tops) {
  if (!h || !tops || min_ms < 0.f || min_ms > 2000.f) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_mfma_i8(h->cu_count, tops, random_operands ? 1 : 0, min_ms);
}

int mmh_probe_hbm_copy(mmh_handle_t h, size_t bytes, float *gbps) {
  if (!h || !gbps || bytes < (1u << 20)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_hbm_copy(bytes, gbps, h->cu_count, 0);
}

int mmh_probe_hbm_read(mmh_handle_t h, size_t bytes, float *gbps) {
  if (!h || !gbps || bytes < (1u << 20)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_hbm_copy(bytes, gbps, h->cu_count, 1);
}

int mmh_probe_lds_read(mmh_handle_t h, int width, float *gbps) {
  if (!h || !gbps || (width != 16 && width != 8 && width != 4 && width != -8)) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  return probe_lds_read(width, gbps, h->cu_count);
}

}  // extern "C"
