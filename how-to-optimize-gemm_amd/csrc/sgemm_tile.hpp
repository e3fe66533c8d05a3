// sgemm_tile.hpp -- shared pieces of the MI355X SGEMM kernels: tile geometry,
// the XCD-aware block -> C-tile map, and the global -> LDS "packing" stage.
//
// Role in the reference's terms (paths relative to /root/reference): this is
// what PackMatrixA/PackMatrixB do for the CPU kernels
// (aarch64/MMult_4x4_12.cpp:41-63) and what the gmem->smem copy with a
// k-major A does for the CUDA kernels (cuda/MMult_cuda_9.cu:43-63): re-lay
// out one K-slice of the A and B panels so the inner kernel reads its
// operands contiguously.  The layout itself is designed for CDNA4's LDS and
// MFMA operand shapes, not translated.
//
// LDS image of one K-slice (BK = 32 deep), per buffer:
//   As[k][m]  k-major ("transposed") A panel, BM floats per k-row, with the
//             16-byte slot index (m/4) XOR-swizzled by swz(k/4);
//   Bs[k][n]  B panel exactly as it lies in memory, BN floats per k-row.
// A wave then fetches, for one k-step of 4, its whole 64-row A fragment set
// and 64-column B fragment set with ONE ds_read_b128 each: lane (i = l&15,
// kq = l>>4) reads As[k0+kq][m0+4i..4i+3] -> the A operands of the four
// 16x16 tiles whose rows are interleaved {m0+4i+t}, t = 0..3; likewise B.
// With a row pitch that is a multiple of 256 B those reads are bank-conflict
// free on gfx950 (ds_read_b128 is served in the 16-lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; MI355X_MICROARCH.md, LDS).
//
// The swizzle: A arrives row-major (k contiguous), so a thread that loads a
// 4(m) x 4(k) block as four float4 along k holds the transposed block in
// registers for free and writes four ds_write_b128, one per k.  The eight
// lanes of a ds_write_b128 service group hold eight different k-chunks c of
// the same four rows; without a swizzle they would all hit the same 4 banks
// (k-rows are 512 B apart).  XOR-ing the slot index with
//   swz(c) = (c & 7) | ((c & 4) << 1)        (bits 0-2 = c, bit 3 = bit 2)
// spreads them over all 32 banks, and -- because a k-step's four k-rows share
// one c, and flipping slot bits {0,1} or {2,3 together} maps each read
// service group onto itself -- leaves the fragment reads conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "ab_build.hpp"

namespace mmh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;       // K-slice depth staged per LDS buffer
constexpr int NXCD = 8;      // XCDs on MI355X; block b is observed on XCD b % 8
constexpr int GROUP_M = 8;   // tile-rows per rasterisation group (L2 reuse)

__device__ __forceinline__ int swz_slot(int c) { return (c & 7) | ((c & 4) << 1); }

// > 64 KiB of dynamic LDS must be opted into, per kernel symbol and per device; done once and
// remembered, so that steady-state launches carry no attribute call (and can be captured into a
// hipGraph after one eager warm-up call).
inline hipError_t opt_in_big_lds(const void *kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  struct Done { const void *kernel; int dev; size_t bytes; };
  static std::mutex mu;
  static std::vector<Done> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
      if (d.kernel == kernel && d.dev == dev && d.bytes >= bytes) return hipSuccess;
  }
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  done.push_back(Done{kernel, dev, bytes});
  return hipSuccess;
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Bijective remap of the 1-D block id so that each XCD (private 4 MiB L2)
// works on a contiguous run of C tiles, then a grouped raster inside the run
// so that co-resident blocks share A row-panels and B column-panels.
// (gm: tile rows per rasterisation group -- GROUP_M everywhere in the product; the tools build varies it for the
// L2-reuse measurements of profiles/r04_notes.md)
__device__ __forceinline__ void block_to_tile_g(int bid, int nblk, int nbm, int nbn, int gm, int &tm, int &tn) {
  const int xcd = bid % NXCD;
  const int local = bid / NXCD;
  const int q = nblk / NXCD, r = nblk % NXCD;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int per_group = gm * nbn;
  // (round 5: the two integer divisions below are ~55 scalar instructions in front of every workgroup's first load; the
  // reference sweep's grids -- 16 x 16 tiles at N = 1024 -- have power-of-two group sizes: shifts, behind a uniform branch)
  int group;
  if ((per_group & (per_group - 1)) == 0) group = logical >> __builtin_ctz(per_group);
  else group = logical / per_group;
  const int first_m = group * gm;
  const int gsize = min(nbm - first_m, gm);
  const int in_group = logical - group * per_group;
  if ((gsize & (gsize - 1)) == 0) {
    tm = first_m + (in_group & (gsize - 1));
    tn = in_group >> __builtin_ctz(gsize);
  } else {
    tn = in_group / gsize;
    tm = first_m + in_group - tn * gsize;
  }
}
// the same map with its two divisions taken unconditionally (rounds 1-4's form).  K1W keeps it: measured with the
// power-of-two fast path above its 128x64 tile ran 1-4 % SLOWER (N = 1152 .. 1408, 2304: two builds side by side in one
// process, tools/ab_lib.py) -- nothing in the map's values or the kernel's registers differs, the VALU loop's code just
// lands elsewhere -- while the MFMA tiles gained 0.4-2.7 % from it.
__device__ __forceinline__ void block_to_tile_g_div(int bid, int nblk, int nbm, int nbn, int gm, int &tm, int &tn) {
  const int xcd = bid % NXCD;
  const int local = bid / NXCD;
  const int q = nblk / NXCD, r = nblk % NXCD;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int per_group = gm * nbn;
  const int group = logical / per_group;
  const int first_m = group * gm;
  const int gsize = min(nbm - first_m, gm);
  const int in_group = logical - group * per_group;
  tm = first_m + in_group % gsize;
  tn = in_group / gsize;
}
__device__ __forceinline__ void block_to_tile(int bid, int nblk, int nbm, int nbn, int &tm, int &tn) {
  block_to_tile_g(bid, nblk, nbm, nbn, GROUP_M, tm, tn);
}

// What a split-K finisher adds to its accumulators before it stores the tile (sgemm_mfma_splitk_kernel):
// `count` dense BM x BN partial tiles, `stride` floats apart, published by `count` arrivals on `*flag`.
struct SplitFix {
  const float *parts = nullptr;
  size_t stride = 0;
  int count = 0;
  int *flag = nullptr;
  int *err = nullptr;
  long long spin_limit = 0;
};

// ---------------------------------------------------------------------------
// Staging registers for one K-slice of a BM x BK A panel and BK x BN B panel,
// THREADS threads.  Each thread owns A_BLKS 4x4 blocks of A and B_VECS float4
// of B.
// ---------------------------------------------------------------------------
// B_HALFSWAP: the B image is read with 8-byte fragments (64x32 wave tiles); odd
// k-rows then store their two 32-float halves swapped (slot bit 3 flipped) so that
// the kq = 0 and kq = 1 lanes of a ds_read_b64 half hit different banks.
// A_HALFSWAP: same for the A image when the wave tile is 32 rows high (8-byte A
// fragments); the half flips with k & 1, i.e. with s & 1 inside a 4x4 block.
// KB: K-slice depth staged per LDS buffer (32, or 128 for the small-tile kernel whose
// one-wave-per-SIMD occupancy needs a longer prefetch distance).
template <int BM, int BN, int THREADS, bool B_HALFSWAP = false, bool A_HALFSWAP = false, int KB = BK>
struct Stage {
  static constexpr int CH = KB / 4;                                   // 4-float k-chunks per row
  static constexpr int A_BLKS = (BM / 4) * CH / THREADS;              // 4x4 blocks per thread
  static constexpr int B_VECS = KB * (BN / 4) / THREADS;              // float4 per thread
  static constexpr int B_ROWS_PER_PASS = THREADS / (BN / 4);
  static_assert(A_BLKS >= 1 && B_VECS >= 1, "tile too small for the block");
  static_assert((BM / 4) * CH == A_BLKS * THREADS && KB * (BN / 4) == B_VECS * THREADS,
                "tile does not divide evenly over the threads");

  f32x4 a[A_BLKS][4];  // a[blk][j] = A[row 4q+j][k 4c..4c+3]
  f32x4 b[B_VECS];

  // Full-tile, 16-byte-aligned fast path.
  __device__ __forceinline__ void load(const float *__restrict__ A, int lda,
                                       const float *__restrict__ B, int ldb, int row0,
                                       int col0, int k0, int tid) {
    const int c = tid % CH;
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk) {
      const int q = tid / CH + blk * (THREADS / CH);
      const float *p = A + (size_t)(row0 + 4 * q) * lda + k0 + 4 * c;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        a[blk][j] = *reinterpret_cast<const f32x4 *>(p + (size_t)j * lda);
    }
    const int cb = tid % (BN / 4);
    const int kr = tid / (BN / 4);
#pragma unroll
    for (int v = 0; v < B_VECS; ++v) {
      const float *p = B + (size_t)(k0 + kr + v * B_ROWS_PER_PASS) * ldb + col0 + 4 * cb;
      b[v] = *reinterpret_cast<const f32x4 *>(p);
    }
  }

  // Same fast path through buffer descriptors (T8): the per-lane byte offsets
  // are loop-invariant VGPRs, the K-slice advance is a scalar soffset, so the
  // steady-state loop carries no 64-bit VALU address arithmetic and each load
  // ships 4 address bytes per lane instead of 8.  rsrc_a covers A from
  // (row0, 0), rsrc_b covers B from (0, col0); offsets must stay < 2^32.
  __device__ __forceinline__ void buf_offsets(int lda, int ldb, int tid, uint32_t (&voff_a)[A_BLKS],
                                              uint32_t &voff_b) const {
    const int c = tid % CH;
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk) {
      const int q = tid / CH + blk * (THREADS / CH);
      voff_a[blk] = (uint32_t)((4 * q) * lda + 4 * c) * 4u;
    }
    voff_b = (uint32_t)((tid / (BN / 4)) * ldb + 4 * (tid % (BN / 4))) * 4u;
  }
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rsrc_a, __amdgpu_buffer_rsrc_t rsrc_b,
                                           const uint32_t (&voff_a)[A_BLKS], uint32_t voff_b,
                                           int lda, int ldb, int k0) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        a[blk][j] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[blk], (k0 + j * lda) * 4, 0));
#pragma unroll
    for (int v = 0; v < B_VECS; ++v)
      b[v] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, voff_b,
                                                       (k0 + v * B_ROWS_PER_PASS) * ldb * 4, 0));
  }

  // LDS-DMA form for B (buffer_load_dwordx4 ... lds): the B image is exactly the
  // memory layout (rows of BN floats, no swizzle, 64x64 wave tiles only), and the
  // thread -> (row, column slot) map above is lane-linear inside a wave, so each
  // wave-instruction drops 1 KiB = 1024/(4 BN) whole k-rows straight into the
  // image: no VGPR round trip, no ds_write.  `bs` = B image of the target buffer.
  // Out-of-range rows (k tail, guarded launches) arrive as zeros like any
  // descriptor-bounded load.
  __device__ __forceinline__ void dma_b(__amdgpu_buffer_rsrc_t rsrc_b, float *bs, uint32_t voff_b,
                                        int ldb, int k0, int wave) const {
    static_assert(!B_HALFSWAP, "the DMA image cannot be swizzled");
    constexpr int ROWS_PER_WAVE = 64 / (BN / 4);
#pragma unroll
    for (int v = 0; v < B_VECS; ++v) {
      float *dst = bs + (wave * ROWS_PER_WAVE + v * B_ROWS_PER_PASS) * BN;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc_b, (__attribute__((address_space(3))) void *)dst, 16, voff_b,
          (k0 + v * B_ROWS_PER_PASS) * ldb * 4, 0, 0);
    }
  }
  __device__ __forceinline__ void load_buf_a(__amdgpu_buffer_rsrc_t rsrc_a, const uint32_t (&voff_a)[A_BLKS],
                                             int lda, int k0) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        a[blk][j] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[blk], (k0 + j * lda) * 4, 0));
  }
  __device__ __forceinline__ void store_a(float *As, int tid) const {
    const int c = tid % CH;
    const int g = A_HALFSWAP ? (c & 7) : swz_slot(c);
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk) {
      const int q = tid / CH + blk * (THREADS / CH);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int slot = A_HALFSWAP ? (q ^ g ^ ((s & 1) << 3)) : (q ^ g);
        f32x4 v = {a[blk][0][s], a[blk][1][s], a[blk][2][s], a[blk][3][s]};
        *reinterpret_cast<f32x4 *>(As + (4 * c + s) * BM + 4 * slot) = v;
      }
    }
  }

  // K tail of the guarded buffer path: in the last, partial K-slice the A loads
  // run past column k into the next row (or the caller's padding); zero those
  // lanes.  (B needs nothing: its rows >= k lie beyond the descriptor's extent
  // and read as 0.)  `krem` = number of valid k in this slice.
  __device__ __forceinline__ void mask_k_tail(int krem, int tid) {
    const int c = tid % CH;
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (4 * c + s >= krem) a[blk][j][s] = 0.0f;
  }

  // Guarded path: any m, n, k, any alignment; out-of-range elements read as 0
  // (a zero product is an exact no-op on an fmaf chain unless the partner is
  // inf/nan, which the fast path would not mask either side of the edge).
  __device__ __forceinline__ void load_edge(const float *__restrict__ A, int lda,
                                            const float *__restrict__ B, int ldb, int row0,
                                            int col0, int k0, int m, int n, int k, int tid) {
    const int c = tid % CH;
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk) {
      const int q = tid / CH + blk * (THREADS / CH);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = row0 + 4 * q + j;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = k0 + 4 * c + s;
          a[blk][j][s] = (row < m && kk < k) ? A[(size_t)row * lda + kk] : 0.0f;
        }
      }
    }
    const int cb = tid % (BN / 4);
    const int kr = tid / (BN / 4);
#pragma unroll
    for (int v = 0; v < B_VECS; ++v) {
      const int kk = k0 + kr + v * B_ROWS_PER_PASS;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int col = col0 + 4 * cb + s;
        b[v][s] = (kk < k && col < n) ? B[(size_t)kk * ldb + col] : 0.0f;
      }
    }
  }

  // Registers -> LDS.  As/Bs point at the destination buffer.
  __device__ __forceinline__ void store(float *As, float *Bs, int tid) const {
    const int c = tid % CH;
    const int g = A_HALFSWAP ? (c & 7) : swz_slot(c);
#pragma unroll
    for (int blk = 0; blk < A_BLKS; ++blk) {
      const int q = tid / CH + blk * (THREADS / CH);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int slot = A_HALFSWAP ? (q ^ g ^ ((s & 1) << 3)) : (q ^ g);
        f32x4 v = {a[blk][0][s], a[blk][1][s], a[blk][2][s], a[blk][3][s]};
        *reinterpret_cast<f32x4 *>(As + (4 * c + s) * BM + 4 * slot) = v;
      }
    }
    const int cb = tid % (BN / 4);
    const int kr = tid / (BN / 4);
#pragma unroll
    for (int v = 0; v < B_VECS; ++v) {
      const int row = kr + v * B_ROWS_PER_PASS;
      const int slot = B_HALFSWAP ? (cb ^ ((row & 1) << 3)) : cb;
      *reinterpret_cast<f32x4 *>(Bs + row * BN + 4 * slot) = b[v];
    }
  }
};

}  // namespace mmh
