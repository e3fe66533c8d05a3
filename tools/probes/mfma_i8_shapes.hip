// mfma_i8_shapes.hip -- sustained MFMA-only rate of the two double-rate int8 shapes on gfx950 and of
// v_mfma_i32_16x16x32_i8 (the CDNA3-era shape BASELINE.json configs[4] names; round 4), constant
// and pseudo-random operands (the power manager decides what random data sustains).  (tools only)
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_i8_shapes.hip -o /tmp/mfma_i8_shapes && /tmp/mfma_i8_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void fill(i32x4 (&a)[4], i32x4 (&b)[4], int rnd, int seed) {
  unsigned x = (unsigned)seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9973u + 12345u;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      x = x * 1664525u + 1013904223u;
      a[i][j] = rnd ? (int)(x ^ (x >> 15)) : 0x01010101;
      x = x * 1664525u + 1013904223u;
      b[i][j] = rnd ? (int)(x ^ (x >> 13)) : 0x02020202;
    }
}

__global__ void __launch_bounds__(256) k16(int *out, int iters, int rnd) {
  i32x4 a[4], b[4];
  fill(a, b, rnd, 1);
  i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
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
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  i32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (s[0] + s[1] + s[2] + s[3] == 123456789) out[0] = s[0];
}

// v_mfma_i32_16x16x32_i8: A and B are 8 bytes per lane (two VGPRs), half the K of the x64 form per instruction
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k16x32(int *out, int iters, int rnd) {
  i32x4 a4[4], b4[4];
  fill(a4, b4, rnd, 1);
  i32x2 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = i32x2{a4[i][0], a4[i][1]};
    b[i] = i32x2{b4[i][0], b4[i][1]};
  }
  i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "v_mfma_i32_16x16x32_i8 %0, %8, %12, %0\n\t"
        "v_mfma_i32_16x16x32_i8 %1, %9, %13, %1\n\t"
        "v_mfma_i32_16x16x32_i8 %2, %10, %14, %2\n\t"
        "v_mfma_i32_16x16x32_i8 %3, %11, %15, %3\n\t"
        "v_mfma_i32_16x16x32_i8 %4, %8, %13, %4\n\t"
        "v_mfma_i32_16x16x32_i8 %5, %9, %14, %5\n\t"
        "v_mfma_i32_16x16x32_i8 %6, %10, %15, %6\n\t"
        "v_mfma_i32_16x16x32_i8 %7, %11, %12, %7"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  i32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (s[0] + s[1] + s[2] + s[3] == 123456789) out[0] = s[0];
}

__global__ void __launch_bounds__(256) k32(int *out, int iters, int rnd) {
  i32x4 a[4], b[4];
  fill(a, b, rnd, 1);
  i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    asm volatile(
        "v_mfma_i32_32x32x32_i8 %0, %4, %8, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %5, %9, %1\n\t"
        "v_mfma_i32_32x32x32_i8 %2, %6, %10, %2\n\t"
        "v_mfma_i32_32x32x32_i8 %3, %7, %11, %3\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %5, %10, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %6, %11, %1\n\t"
        "v_mfma_i32_32x32x32_i8 %2, %7, %8, %2\n\t"
        "v_mfma_i32_32x32x32_i8 %3, %4, %9, %3"
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  i32x16 s = c0 + c1 + c2 + c3;
  int t = 0;
  for (int i = 0; i < 16; ++i) t += s[i];
  if (t == 123456789) out[0] = t;
}

int main() {
  int *d;
  hipMalloc(&d, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 512, iters = 20000;
  for (int shape = 0; shape < 3; ++shape)
    for (int rnd = 0; rnd < 2; ++rnd) {
      float ms = 0, total = 0;
      auto launch = [&](int n) {
        if (shape == 0) hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, n, rnd);
        else if (shape == 1) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, d, n, rnd);
        else hipLaunchKernelGGL(k16x32, dim3(blocks), dim3(256), 0, 0, d, n, rnd);
      };
      launch(2000);
      do {   // ~100 ms back to back: the sustained state
        hipEventRecord(e0, 0);
        launch(iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        total += ms;
      } while (total < 100.f);
      const double macs = shape == 0 ? 16.0 * 16 * 64 : shape == 1 ? 32.0 * 32 * 32 : 16.0 * 16 * 32;
      const double ops = (double)blocks * 4 * iters * 8.0 * 2.0 * macs;
      printf("%s, %s operands: %7.1f TOPS (last launch %.2f ms)\n", shape == 0 ? "v_mfma_i32_16x16x64_i8" : shape == 1 ? "v_mfma_i32_32x32x32_i8" : "v_mfma_i32_16x16x32_i8",
             rnd ? "random  " : "constant", ops / (ms * 1e-3) / 1e12, ms);
    }
  return 0;
}
