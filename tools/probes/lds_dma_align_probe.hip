// lds_dma_align_probe.hip -- empirical semantics of `buffer_load_dwordx4 ... lds` (LDS-DMA) on gfx950, the
// questions the guarded LDS-DMA tiles (csrc/sgemm_dma.hpp, EDGE) rest on:
//  (1) a 16-byte piece whose SOURCE is only 4-byte aligned (odd leading dimension / base): right bytes in LDS?
//  (2) a piece that straddles the descriptor's extent: in-range dwords arrive, the others are ZERO in LDS
//      (not stale -- the LDS is pre-filled with a sentinel)?
//  (3) a piece entirely out of range: zeros?
// Build: hipcc --offload-arch=gfx950 -O2 lds_dma_align_probe.hip -o lds_dma_align_probe.x ; prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(const int *src, int *out, int nrec_bytes, int shift_dwords) {
  __shared__ __attribute__((aligned(16))) int lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0x5e471e1;   // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nrec_bytes, 0x00020000);
  // lane L: 16 bytes from byte offset 16 L + 4 shift -> LDS bytes 16 L
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16,
                                           (unsigned)(16 * threadIdx.x + 4 * shift_dwords), 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  const int N = 1024;
  int *h = (int *)malloc(N * 4), *src, *out, o[256];
  for (int i = 0; i < N; ++i) h[i] = 100000 + i;
  hipMalloc(&src, N * 4); hipMalloc(&out, 256 * 4);
  hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
  int fails = 0;
  for (int shift = 0; shift < 4; ++shift) {                       // (1) alignment
    probe<<<1, 64>>>(src, out, N * 4, shift);
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += o[i] != 100000 + i + shift;
    printf("%s source shifted by %d dword(s): %d wrong of 256 (e.g. lds[5] = %d, want %d)\n", bad ? "FAIL" : "PASS", shift, bad,
           o[5], 100005 + shift);
    fails += bad != 0;
  }
  for (int shift = 0; shift < 2; ++shift) {                       // (2) + (3): extent ends inside lane 10's piece
    const int visible = 42;                                       // dwords 0..41 visible
    probe<<<1, 64>>>(src, out, visible * 4, shift);
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0, stale = 0;
    for (int i = 0; i < 256; ++i) {
      const int srci = i + shift;
      const int want = srci < visible ? 100000 + srci : 0;
      bad += o[i] != want;
      stale += o[i] == 0x5e471e1;
    }
    printf("%s extent at dword %d, shift %d: %d wrong of 256, %d sentinels left (lds[40..47] = %d %d %d %d %d %d %d %d)\n",
           bad ? "FAIL" : "PASS", visible, shift, bad, stale, o[40], o[41], o[42], o[43], o[44], o[45], o[46], o[47]);
    fails += bad != 0;
  }
  printf(fails ? "lds_dma_align_probe: %d check(s) FAILED\n" : "lds_dma_align_probe: all checks passed (%d)\n", fails);
  return 0;
}
