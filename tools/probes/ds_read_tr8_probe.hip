#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(const int *addr_in, uint32_t *out, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(i & 255);
  __syncthreads();
  const int a = addr_in[threadIdx.x];
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(lds + a));
  out[2 * threadIdx.x] = (uint32_t)v[0];
  out[2 * threadIdx.x + 1] = (uint32_t)v[1];
  // also record the high byte of address block to disambiguate >256
  out[128 + threadIdx.x] = (uint32_t)a;
}
int main() {
  int h_addr[64]; uint32_t h_out[192];
  int *d_addr; uint32_t *d_out;
  hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
  for (int mode = 0; mode < 4; ++mode) {
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) h_addr[l] = 0;                 // all lanes same address
      if (mode == 1) h_addr[l] = 8 * l;             // contiguous 8 B per lane
      if (mode == 2) h_addr[l] = 16 * (l & 7) + 128 * (l >> 3);   // lanes 0-7: one 128-B row? 
      if (mode == 3) h_addr[l] = 256 * (l & 15) + 8 * (l >> 4);   // per-lane row pitch 256
    }
    hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out, mode);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d addr %4d :", l, h_addr[l]);
      for (int j = 0; j < 8; ++j) printf(" %3u", (h_out[2 * l + j / 4] >> (8 * (j % 4))) & 255);
      printf("\n");
    }
  }
  return 0;
}
