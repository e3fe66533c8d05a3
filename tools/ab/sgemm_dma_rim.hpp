// sgemm_dma_rim.hpp -- round 3's rim for the K2L tiles (tools build only; moved out of csrc/sgemm_dma.hpp in round 5: the
// product's kernel headers hold what ships).  Included by csrc/launch_dma.hip under -DMMH_AB_BUILD (-I tools/ab).
#pragma once
#include "sgemm_dma.hpp"

namespace mmh {

// ---- the rim (opt-in: MMH_OPT_RIM; measured, it does not pay -- profiles/r03_notes.md section 6) ---------------------
// A shape a few elements past a tile boundary (N = 1025: `m % 64`, `n % 64` small) pays a whole extra row AND column
// of tiles for one element each way -- 13 % more tile work at N = 1025 and, worse, a second tile on 33 of 256 CUs
// (80 TFLOP/s against 107-117 at N = 1024).  sgemm_mfma_dma_rim_kernel runs the TRIMMED problem (m0 = m - m % 64,
// n0 = n - n % 64) on the MFMA tiles and the rim -- the strip of columns n0 .. n-1 (all rows, corner included) and the
// strip of rows m0 .. m-1 (columns below n0) -- on the vector ALU in extra workgroups of the same launch: one lane per
// C element, one `v_fma_f32` chain over ascending k.  Same chain as the MFMA (which is what makes K1 and K2 agree bit
// for bit), so the same bits (tests/test_gpu_round3.py).
//
// A chain is k dependent FMAs -- 4 cycles each, a quarter of what a tile's K loop takes -- but every link needs two
// operands from memory, and what ONE wave gets out of the memory system is (bytes in flight) / latency.  A rim unit is
// SIXTEEN elements, one wave per workgroup (the other waves leave at once); the workgroup's whole LDS allocation --
// it is there anyway, the launch is sized for the tiles -- is that wave's prefetch ring: slots of 128 links, filled
// by LDS-DMA (no registers), all but one in flight under a counted `s_waitcnt vmcnt`, exactly the tiles' own scheme.
// Per element and link the lane needs one value of its own (x) and one the whole wave shares (u):
//   right strip, unit (column c, 16 rows):  x = A[row][k]  (a DMA instruction = two rows x 512 B, whole lines; the
//                                           16-byte pieces XOR-swizzled by the row so that the sixteen readers hit
//                                           sixteen bank groups),  u = B[k][n0 + c]  (a dword per lane = 64 k);
//   bottom strip, unit (row r, 16 columns): x = B[k][col]  (sixteen k-rows of 64 B per DMA instruction;
//                                           ds_read_b32 per link),   u = A[m0 + r][k].
// Descriptor extents turn rows past the end into zeros; links past k exist only in the last slot, whose copy of
// the chain drops them.
// What was measured (N = 1025, tools/rim_ab.py with the A/B library's "rim alone" / "tiles alone" switches): units of
// 64 elements, 128 links in flight: 26-33 ns per link, the launch 34 us (63 TFLOP/s); units of 16 elements, 512 links in
// flight: the rim ALONE 13 us, the 256 tiles ALONE 20 us (= N = 1024) -- and together 28-29 us, whichever comes first in
// dispatch order, whatever `s_setprio` either side runs at; the plain launch of 17 x 17 edge tiles takes 27 us.  A rim
// wave beside a tile's four waves on a CU costs that tile more than the second, thin edge tile did.  (Also measured on
// the way: interleaving rim units with the tiles in dispatch order, or letting the chain's register blocks unroll --
// 312 registers, one workgroup per CU -- each cost the TILES 60 % at one tile per CU.)
template <int LDS_FLOATS>
__device__ __forceinline__ void rim_body(float *lds, int unit, int m, int n, int k, const float *__restrict__ A,
                                         int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc,
                                         bool accumulate, int m0, int n0) {
  if (threadIdx.x >= 64) return;
  constexpr int E = 16, KC = 128, SUB = 32;      // elements per unit, links per slot, links per register block
  constexpr int SLOT = E * KC + KC;              // floats per slot: x image + u
  constexpr int PER = E * KC / 256 + KC / 64;    // DMA instructions per slot: 8 of 1 KiB + 2 of 256 B
  constexpr int R = LDS_FLOATS / SLOT < 7 ? LDS_FLOATS / SLOT : 7;   // ring slots (64x64 tiles: 5, 128x64: 7)
  static_assert(R >= 3 && R * SLOT <= LDS_FLOATS, "the rim's ring lives in the tile's LDS allocation");
  static_assert((R - 1) * PER <= 63, "vmcnt is a 6-bit counter");
  const int lane = threadIdx.x, e = lane & (E - 1);
  const int rn = n - n0;
  const int row_blocks = (m + E - 1) / E, col_blocks = n0 / E;
  const bool right = unit < rn * row_blocks;
  int i, j;
  bool valid;
  const float *xbase, *ubase;
  uint32_t ext_x, ext_u, voff_x[8], voff_u, sx, su, gx;
  if (right) {
    const int c = unit / row_blocks, row0 = (unit % row_blocks) * E;
    i = row0 + e;
    j = n0 + c;
    valid = lane < E && i < m;
    xbase = A + (size_t)row0 * lda;
    ext_x = (uint32_t)(((min(E, m - row0) - 1) * lda + k) * 4);
#pragma unroll
    for (int g = 0; g < 8; ++g) {   // piece g: rows 2 g, 2 g + 1; lane -> (row, physical 16-byte slot lane % 32)
      const int row = 2 * g + (lane >> 5);
      voff_x[g] = (uint32_t)(row * lda + 4 * ((lane & 31) ^ row)) * 4u;
    }
    sx = 4u;
    gx = 0u;
    ubase = B + j;
    ext_u = (uint32_t)(((k - 1) * ldb + 1) * 4);
    voff_u = (uint32_t)(lane * ldb) * 4u;
    su = (uint32_t)ldb * 4u;
  } else {
    const int u2 = unit - rn * row_blocks, col0 = (u2 % col_blocks) * E;
    i = m0 + u2 / col_blocks;
    j = col0 + e;
    valid = lane < E;
    xbase = B + col0;
    ext_x = (uint32_t)(((k - 1) * ldb + E) * 4);
#pragma unroll
    for (int g = 0; g < 8; ++g) voff_x[g] = (uint32_t)((lane >> 2) * ldb + 4 * (lane & 3)) * 4u;
    sx = (uint32_t)ldb * 4u;
    gx = 16u * sx;
    ubase = A + (size_t)i * lda;
    ext_u = (uint32_t)(k * 4);
    voff_u = (uint32_t)lane * 4u;
    su = 4u;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xbase), 0, ext_x, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ubase), 0, ext_u, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xbase), 0, 0, 0x00020000);
  float *c_ptr = C + (size_t)i * ldc + j;
  float acc = (accumulate && valid) ? *c_ptr : 0.0f;
  const int nchunks = (k + KC - 1) / KC;
  auto issue = [&](int c) {   // chunk c into slot c % R (past the end: the same instructions against an empty descriptor)
    float *slot = lds + (c % R) * SLOT;
    const bool live = c < nchunks;
    const uint32_t kc = (uint32_t)(c * KC);
#pragma unroll
    for (int g = 0; g < 8; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? rsrc_x : null_x, (__attribute__((address_space(3))) void *)(slot + 256 * g),
                                               16, voff_x[g], kc * sx + (uint32_t)g * gx, 0, 0);
#pragma unroll
    for (int h = 0; h < KC / 64; ++h)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? rsrc_u : null_x,
                                               (__attribute__((address_space(3))) void *)(slot + E * KC + 64 * h), 4, voff_u,
                                               (kc + 64u * h) * su, 0, 0);
  };
  // one chunk: 128 links of the chain in four register blocks.  Only the LAST chunk can hold links past k (what lies
  // there is the next row, or the caller's padding): every other chunk runs the bare chain, one dependent v_fma_f32
  // per link.
  auto chunk = [&](int c, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot about to be refilled has been read
    issue(c + R - 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * PER) : "memory");   // chunk c has landed
    const float *slot = lds + (c % R) * SLOT;
    // (a rolled loop: unrolled, hipcc reads all four blocks' operands first -- 256 registers, one wave per SIMD, and
    // the tiles of this kernel then run one workgroup per CU instead of three)
#pragma unroll 1
    for (int sb = 0; sb < KC / SUB; ++sb) {
      const int kend = k - c * KC - sb * SUB;   // links of this block that exist
      float x[SUB], u[SUB];
#pragma unroll
      for (int g = 0; g < SUB / 4; ++g) {
        const f32x4 uv = *reinterpret_cast<const f32x4 *>(slot + E * KC + SUB * sb + 4 * g);
        u[4 * g] = uv[0]; u[4 * g + 1] = uv[1]; u[4 * g + 2] = uv[2]; u[4 * g + 3] = uv[3];
      }
      if (right) {
#pragma unroll
        for (int g = 0; g < SUB / 4; ++g) {
          const f32x4 xv = *reinterpret_cast<const f32x4 *>(slot + KC * e + 4 * ((SUB / 4 * sb + g) ^ e));
          x[4 * g] = xv[0]; x[4 * g + 1] = xv[1]; x[4 * g + 2] = xv[2]; x[4 * g + 3] = xv[3];
        }
      } else {
#pragma unroll
        for (int q = 0; q < SUB; ++q) x[q] = slot[E * (SUB * sb + q) + e];
      }
      // fmaf(a, b, acc) with a from A and b from B either way (the product is commutative bit for bit)
#pragma unroll
      for (int q = 0; q < SUB; ++q) {
        const float next = __builtin_fmaf(x[q], u[q], acc);
        if constexpr (LAST) acc = q < kend ? next : acc;
        else acc = next;
      }
    }
  };
  for (int c = 0; c < R - 1; ++c) issue(c);
  for (int c = 0; c < nchunks - 1; ++c) chunk(c, std::false_type{});
  chunk(nchunks - 1, std::true_type{});
  if (valid) *c_ptr = acc;
}

// Workgroups 0 .. nbm*nbn-1: the tiles of the trimmed problem (m0 x n0, as sgemm_mfma_dma_kernel), placed by the
// dispatcher exactly as a plain launch of the trimmed shape would be; the rest: the rim's units, which take the
// workgroup slots the tiles leave free.  The launcher only builds such a launch when tiles AND units are all
// resident from the start (measured, profiles/r03_notes.md section 6: a unit's chain is as long as a tile's K loop and
// cannot be split, so behind a second round of tiles it is the launch's tail; in front of the tiles, or interleaved
// with them, it unbalances which CU gets which tile -- with about one tile per CU that costs more than the rim saves).
template <int BM, int BN, int KB, int WTM, int WTN, int NBUF, bool EDGE = false>
__global__ void __launch_bounds__((BM / (16 * WTM)) * (BN / (16 * WTN)) * 64)
sgemm_mfma_dma_rim_kernel(int m0, int n0, int k, const float *__restrict__ A, int lda, const float *__restrict__ B,
                          int ldb, float *__restrict__ C, int ldc, int accumulate, int nbm, int nbn, int m, int n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tiles = nbm * nbn;
  // tools/rim_ab.py: the rim alone (accumulate bit 1) / the trimmed tiles alone (bit 2) -- wrong results
  if ((accumulate & 2) && (int)blockIdx.x < tiles) return;
  if ((accumulate & 4) && (int)blockIdx.x >= tiles) return;
  accumulate &= 1;
  if ((int)blockIdx.x >= tiles) {
    rim_body<DmaTile<BM, BN, KB, WTM, WTN, NBUF>::LDS_BYTES / 4>(lds, (int)blockIdx.x - tiles, m, n, k, A, lda, B, ldb, C, ldc,
                                                                 accumulate != 0, m0, n0);
    return;
  }
  int tm, tn;
  block_to_tile(blockIdx.x, tiles, nbm, nbn, tm, tn);
  DmaSegment<BM, BN, KB, WTM, WTN, NBUF, false, EDGE>::run(lds, m0, n0, k, A, lda, B, ldb, C, ldc, tm, tn, 0,
                                                           (k + KB - 1) / KB, accumulate != 0);
}

}  // namespace mmh
