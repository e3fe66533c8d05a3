// hbm_patterns.hip -- which access pattern streams HBM fastest on this part?  (tools only)
// hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_patterns.hip -o /tmp/hbm_patterns && /tmp/hbm_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT, int U>
__global__ void __launch_bounds__(256) copy_gridstride(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = NT ? __builtin_nontemporal_load(s + i + j * stride) : s[i + j * stride];
#pragma unroll
    for (int j = 0; j < U; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + i + j * stride); else d[i + j * stride] = v[j]; }
  }
}
// each block owns one contiguous chunk of 256 * U float4 (U * 4 KiB); one trip, then exits
template <bool NT, int U>
__global__ void __launch_bounds__(256) copy_blockchunk(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int j = 0; j < U; ++j) v[j] = NT ? __builtin_nontemporal_load(s + base + j * 256) : s[base + j * 256];
#pragma unroll
  for (int j = 0; j < U; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + base + j * 256); else d[base + j * 256] = v[j]; }
}
template <int U>
__global__ void __launch_bounds__(256) read_blockchunk(const f4 *__restrict__ s, float *__restrict__ out, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 a = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < U; ++j) a += s[base + j * 256];
  if (a[0] + a[1] + a[2] + a[3] == 12345.6789f) out[0] = a[0];
}
template <int U>
__global__ void __launch_bounds__(256) read_gridstride(const f4 *__restrict__ s, float *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  f4 a = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = s[i + j * stride];
#pragma unroll
    for (int j = 0; j < U; ++j) a += v[j];
  }
  if (a[0] + a[1] + a[2] + a[3] == 12345.6789f) out[0] = a[0];
}
__global__ void __launch_bounds__(256) fill_blockchunk(f4 *__restrict__ d, size_t n) {
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) d[base + j * 256] = f4{1.f, 2.f, 3.f, 4.f};
}

int main() {
  const size_t bytes = 1ull << 30, n = bytes / 16;
  f4 *s, *d;
  CK(hipMalloc(&s, bytes));
  CK(hipMalloc(&d, bytes));
  CK(hipMemset(s, 1, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, double moved, auto &&launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.0f GB/s\n", name, moved * 10 / (ms * 1e-3) / 1e9);
  };
  for (int blocks : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "copy grid-stride U4 plain, %d blocks", blocks);
    run(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_gridstride<false, 4>), dim3(blocks), dim3(256), 0, 0, s, d, n); });
    snprintf(nm, sizeof nm, "copy grid-stride U4 nt, %d blocks", blocks);
    run(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_gridstride<true, 4>), dim3(blocks), dim3(256), 0, 0, s, d, n); });
  }
  run("copy block-chunk U4 plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_blockchunk<false, 4>), dim3(n / 1024), dim3(256), 0, 0, s, d, n); });
  run("copy block-chunk U4 nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_blockchunk<true, 4>), dim3(n / 1024), dim3(256), 0, 0, s, d, n); });
  run("copy block-chunk U8 plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_blockchunk<false, 8>), dim3(n / 2048), dim3(256), 0, 0, s, d, n); });
  run("copy block-chunk U8 nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_blockchunk<true, 8>), dim3(n / 2048), dim3(256), 0, 0, s, d, n); });
  run("copy block-chunk U2 plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_blockchunk<false, 2>), dim3(n / 512), dim3(256), 0, 0, s, d, n); });
  run("copy block-chunk U1 plain", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_blockchunk<false, 1>), dim3(n / 256), dim3(256), 0, 0, s, d, n); });
  run("hipMemcpyDtoD", 2.0 * bytes, [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
  run("read block-chunk U8", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_blockchunk<8>), dim3(n / 2048), dim3(256), 0, 0, s, (float *)d, n); });
  run("read block-chunk U4", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_blockchunk<4>), dim3(n / 1024), dim3(256), 0, 0, s, (float *)d, n); });
  run("read block-chunk U16", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_blockchunk<16>), dim3(n / 4096), dim3(256), 0, 0, s, (float *)d, n); });
  run("read grid-stride U8, 2048 blocks", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_gridstride<8>), dim3(2048), dim3(256), 0, 0, s, (float *)d, n); });
  run("read grid-stride U8, 8192 blocks", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_gridstride<8>), dim3(8192), dim3(256), 0, 0, s, (float *)d, n); });
  run("fill block-chunk U8", 1.0 * bytes, [&] { hipLaunchKernelGGL(fill_blockchunk, dim3(n / 2048), dim3(256), 0, 0, d, n); });
  run("hipMemsetAsync", 1.0 * bytes, [&] { hipMemsetAsync(d, 0, bytes, 0); });
  return 0;
}
