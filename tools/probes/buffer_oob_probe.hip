// buffer_oob_probe.hip -- empirical semantics of raw buffer loads on gfx950:
//  (1) a dwordx4 load straddling num_records: which dwords come back?
//  (2) a dwordx4 load at a 4-byte-aligned (not 16-byte-aligned) offset: works?
//  (3) dwordx4 buffer STORE straddling num_records: which dwords are written?
// Build: hipcc --offload-arch=gfx950 -O2 buffer_oob_probe.hip -o buffer_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int *src, int *out, int *dst, int nrec_bytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nrec_bytes, 0x00020000);
  const int t = threadIdx.x;
  // thread t loads 16 bytes at byte offset 4*t  (t=0..15)
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 4 * t, 0, 0);
  for (int i = 0; i < 4; ++i) out[4 * t + i] = v[i];
  __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, nrec_bytes, 0x00020000);
  if (t == 0) {
    i32x4 s = {101, 102, 103, 104};
    __builtin_amdgcn_raw_buffer_store_b128(s, w, nrec_bytes - 8, 0, 0);   // straddles: 2 dwords in, 2 out
  }
}
int main() {
  int h[32], *src, *out, *dst;
  for (int i = 0; i < 32; ++i) h[i] = 1000 + i;
  hipMalloc(&src, sizeof(h)); hipMalloc(&out, 64 * 4); hipMalloc(&dst, sizeof(h));
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(dst, 0, sizeof(h));
  const int nrec = 10 * 4;   // 10 dwords visible
  probe<<<1, 16>>>(src, out, dst, nrec);
  int o[64], d[32];
  hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
  hipMemcpy(d, dst, sizeof(d), hipMemcpyDeviceToHost);
  printf("num_records = %d bytes (dwords 0..9 visible)\n", nrec);
  for (int t = 0; t < 16; ++t) printf("load @%2d: %d %d %d %d\n", 4 * t, o[4*t], o[4*t+1], o[4*t+2], o[4*t+3]);
  printf("store straddle: dst[8..11] = %d %d %d %d\n", d[8], d[9], d[10], d[11]);
  return 0;
}
