// igemm.hip -- BASELINE.json config 5 and the callers either side of it: mmh_igemm_s8 (int8 x int8 -> int32 on
// v_mfma_i32_16x16x64_i8), mmh_quantize_sym_s8 and mmh_qgemm_f32 (symmetric per-tensor quantisation, dequantisation
// fused into the GEMM's epilogue).  The reference holds no int8 code (README.md:71-85 is prose): parity unpinned.
// Part of libmmult_hip.so (see internal.hpp).
#include <algorithm>

#include "internal.hpp"
#include "igemm_s8.hpp"
#include "igemm_s8_pp.hpp"
#ifdef MMH_AB_BUILD
#include "igemm_s8_k3.hpp"   // tools/ab/: K3 and the packed-B kernel K3d (modes 1, 3, 4, 10..13)
#endif
#include "quant_s8.hpp"

using namespace mmh;

// K3p's launch form.  Persistent (one workgroup per CU walking the tiles) saves a tile's launch and prologue: +1.3 / +4.5 /
// +1.7 % at 8192 x 8192 x 1024 / x 2048 and 5120^3; one workgroup per tile lets the dispatcher hand the next tile to whichever
// CU is free first -- tiles of a long K finish up to 20 us apart (tools/i8_timeline.py) -- and is 1.5 % ahead at 8192^3;
// level in between (profiles/r06_i8_persist_ab.txt).  Mode 8 / 9 force one or the other.
static int pp_grid_cap(const mmh_context *h, int mode, int k, int cus) {
  if (mode == 9) return 0;
  if (h->i8_grid_cap > 0) return h->i8_grid_cap;   // test hook: a few persistent workgroups walk a small shape's tiles
  if (mode == 8) return cus;
  return k <= 5120 ? cus : 0;
}

// tools build: the rungs that left the product (tools/ab/igemm_s8_k3.hpp).  Returns 1 when `mode` is not one of theirs.
static int launch_ab_modes(mmh_context *h, int mode, int m, int n, int k, const int8_t *A, int lda, const int8_t *B, int ldb, int32_t *C,
                           int ldc, int acc, hipStream_t s) {
#ifdef MMH_AB_BUILD
  if (mode == 1 || mode == 3 || mode == 4 || (mode >= 10 && mode <= 13)) {
    int8_t *bt = nullptr;
    if (igemm_s8_needs_pack(mode, A, lda, B, ldb, k) && h->bt.reserve(igemm_s8_pack_bytes(n, k)) == MMH_OK)
      bt = static_cast<int8_t *>(h->bt.p);
    const hipError_t e = launch_igemm_s8_ab(m, n, k, A, lda, B, ldb, C, ldc, acc, s, bt, mode);
    if (e == hipErrorNotSupported) return 1;   // an operand those kernels cannot take: the product's path
    HIP_TRY(e);
    return MMH_OK;
  }
#else
  (void)h; (void)mode; (void)m; (void)n; (void)k; (void)A; (void)lda; (void)B; (void)ldb; (void)C; (void)ldc; (void)acc; (void)s;
#endif
  return 1;
}

extern "C" {

#ifdef MMH_DMA_TIMELINE
// timeline build only: where K3p's waves 0 and 4 write their per-tile stamps (2 x 16 x 8 uint64 per workgroup; NULL = off)
int mmh_ab_set_stamps_i8(mmh_handle_t h, void *stamps) {
  if (!h) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamps), &stamps, sizeof(void *)));
  return MMH_OK;
}
#endif

int mmh_igemm_s8(mmh_handle_t h, int m, int n, int k, const int8_t *dA, int lda, const int8_t *dB,
                 int ldb, int32_t *dC, int ldc, int accumulate, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k == 0) {
    if (!accumulate)
      HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * 4, 0, (size_t)n * 4, (size_t)m, s));
    return MMH_OK;
  }
  // K3p (igemm_s8_pp.hpp): the 256x256 in-place tile with its wave groups in ping-pong, persistent over the tiles --
  // what mode 0 runs from one tile per CU up.  Forced: 8 (as mode 0 launches it), 9 (one workgroup per tile: round 5's
  // launch form, the A/B switch for the persistent loop), 7 (the same kernel on v_mfma_i32_16x16x32_i8, the instruction
  // BASELINE.json configs[4] names: same bits, half the pipe's rate); 6 stays the lockstep K3t kernel.
  const int cus_ = h->cu_count > 0 ? h->cu_count : 256;
  if (igemm_s8_inplace_ok(dA, lda, dB, ldb, k) &&
      (h->igemm_mode == 7 || h->igemm_mode == 8 || h->igemm_mode == 9 ||
       (h->igemm_mode == 0 && igemm_s8_big_tile(m, n, cus_)))) {
    if (h->igemm_mode == 7) HIP_TRY(launch_igemm_s8_pp<32>(m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate ? 1 : 0, s, pp_grid_cap(h, 0, k, cus_)));
    else HIP_TRY(launch_igemm_s8_pp<64>(m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate ? 1 : 0, s, pp_grid_cap(h, h->igemm_mode, k, cus_)));
    return MMH_OK;
  }
  // Default mode: operands the in-place kernel cannot take as they are (an odd leading dimension, a
  // base that is not dword-aligned) are first copied into dense dword-aligned workspace images -- one
  // pass over m*k / k*n bytes, against m*n*k MACs -- instead of falling back to the slow kernels.
  if (h->igemm_mode == 0 && !igemm_s8_inplace_ok(dA, lda, dB, ldb, k)) {
    const bool a_ok = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(dA) & 3) == 0);
    const bool b_ok = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(dB) & 3) == 0);
    const int ka = a_ok ? lda : (k + 15) & ~15, nb = b_ok ? ldb : (n + 15) & ~15;
    const int8_t *sa = dA, *sb = dB;
    if (!a_ok) {
      if ((rc = h->qa.reserve((size_t)m * ka)) != MMH_OK) return rc;
      HIP_TRY(hipMemcpy2DAsync(h->qa.p, (size_t)ka, dA, (size_t)lda, (size_t)k, (size_t)m, hipMemcpyDeviceToDevice, s));
      sa = static_cast<const int8_t *>(h->qa.p);
    }
    if (!b_ok) {
      if ((rc = h->qb.reserve((size_t)k * nb)) != MMH_OK) return rc;
      HIP_TRY(hipMemcpy2DAsync(h->qb.p, (size_t)nb, dB, (size_t)ldb, (size_t)n, (size_t)k, hipMemcpyDeviceToDevice, s));
      sb = static_cast<const int8_t *>(h->qb.p);
    }
    if (igemm_s8_inplace_ok(sa, ka, sb, nb, k)) {
      HIP_TRY(launch_igemm_s8(m, n, k, sa, ka, sb, nb, dC, ldc, accumulate ? 1 : 0, s, 0, cus_));
      return MMH_OK;
    }
    // (operands beyond the descriptors' 2 GiB window: the correctness-first kernel below)
  }
  if ((rc = launch_ab_modes(h, h->igemm_mode, m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate ? 1 : 0, s)) <= 0) return rc;
  HIP_TRY(launch_igemm_s8(m, n, k, dA, lda, dB, ldb, dC, ldc, accumulate ? 1 : 0, s, h->igemm_mode, cus_));
  return MMH_OK;
}

int mmh_quantize_sym_s8(mmh_handle_t h, int rows, int cols, const float *dX, int ldx, int8_t *dQ,
                        int ldq, float *d_scale, void *stream) {
  if (!h || rows < 0 || cols < 0) return MMH_ERR_INVALID_ARG;
  if (rows == 0 || cols == 0) return MMH_OK;
  if (!dX || !dQ || !d_scale || ldx < cols || ldq < cols) return MMH_ERR_INVALID_ARG;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // qs: [0, 2 AMAX_WORDS) abs-max words of a quantised GEMM's A and B, then its two scales, then the
  // abs-max words of stand-alone calls
  constexpr size_t qs_words = 4 * AMAX_WORDS + 16;
  int rc = h->qs.reserve(qs_words * sizeof(unsigned));
  if (rc != MMH_OK) return rc;
  unsigned *amax = static_cast<unsigned *>(h->qs.p) + 2 * AMAX_WORDS + 16;
  HIP_TRY(hipMemsetAsync(amax, 0, 2 * AMAX_WORDS * sizeof(unsigned), s));
  const QuantTensor t{dX, rows, cols, ldx, dQ, ldq}, none{nullptr, 0, 0, 0, nullptr, 0};
  const bool v_in = quant_vec_ok(t, false), v_out = quant_vec_ok(t, true);
  hipLaunchKernelGGL(absmax_kernel, dim3(quant_grid(t, v_in), 1), dim3(256), 0, s, t, none, v_in ? 1 : 0, 0, amax);
  hipLaunchKernelGGL(quantize_kernel, dim3(quant_grid(t, v_out), 1), dim3(256), 0, s, t, none, v_out ? 1 : 0, 0,
                     amax, d_scale);
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

int mmh_qgemm_f32(mmh_handle_t h, int m, int n, int k, const float *dA, int lda, const float *dB,
                  int ldb, float *dC, int ldc, void *stream) {
  if (!h) return MMH_ERR_INVALID_ARG;
  int rc = check_gemm_args(m, n, k, dA, lda, dB, ldb, dC, ldc);
  if (rc != MMH_OK) return rc;
  if (m == 0 || n == 0) return MMH_OK;
  ENTER(h);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k == 0) {
    HIP_TRY(hipMemset2DAsync(dC, (size_t)ldc * 4, 0, (size_t)n * 4, (size_t)m, s));
    return MMH_OK;
  }
  // dense, 16-byte-friendly workspace images: int8 A (m x ka), int8 B (k x nb), int32 C (m x nb)
  const int ka = (k + 15) & ~15, nb = (n + 3) & ~3;
  if ((rc = h->qa.reserve((size_t)m * ka)) != MMH_OK) return rc;
  if ((rc = h->qb.reserve((size_t)k * nb)) != MMH_OK) return rc;
  constexpr size_t qs_words = 4 * AMAX_WORDS + 16;
  if ((rc = h->qs.reserve(qs_words * sizeof(unsigned))) != MMH_OK) return rc;
  int8_t *qa = static_cast<int8_t *>(h->qa.p), *qb = static_cast<int8_t *>(h->qb.p);
  unsigned *amax = static_cast<unsigned *>(h->qs.p);                           // A's words, then B's
  float *scales = reinterpret_cast<float *>(amax + 2 * AMAX_WORDS);       // [0] A, [1] B
  HIP_TRY(hipMemsetAsync(amax, 0, 2 * AMAX_WORDS * sizeof(unsigned), s));
  // A and B share one abs-max launch and one quantisation launch (blockIdx.y picks the tensor)
  const QuantTensor ta{dA, m, k, lda, qa, ka}, tb{dB, k, n, ldb, qb, nb};
  const bool va_in = quant_vec_ok(ta, false), vb_in = quant_vec_ok(tb, false);
  const bool va_out = quant_vec_ok(ta, true), vb_out = quant_vec_ok(tb, true);
  const dim3 gmax(std::max(quant_grid(ta, va_in), quant_grid(tb, vb_in)), 2);
  const dim3 g(std::max(quant_grid(ta, va_out), quant_grid(tb, vb_out)), 2);
  hipLaunchKernelGGL(absmax_kernel, gmax, dim3(256), 0, s, ta, tb, va_in ? 1 : 0, vb_in ? 1 : 0, amax);
  hipLaunchKernelGGL(quantize_kernel, g, dim3(256), 0, s, ta, tb, va_out ? 1 : 0, vb_out ? 1 : 0, amax, scales);
  const int cus = h->cu_count > 0 ? h->cu_count : 256;
  if (h->igemm_mode == 0 && igemm_s8_inplace_ok(qa, ka, qb, nb, k)) {
    // the int8 GEMM dequantises in its epilogue: no int32 image of C at all
    if (igemm_s8_big_tile(m, n, cus))
      HIP_TRY(launch_igemm_s8_pp<64>(m, n, k, qa, ka, qb, nb, reinterpret_cast<int32_t *>(dC), ldc, 0, s, pp_grid_cap(h, 0, k, cus), scales));
    else
      HIP_TRY(launch_igemm_s8_dequant(m, n, k, qa, ka, qb, nb, dC, ldc, scales, s, cus));
    return MMH_OK;
  }
  // two-pass form (A/B modes of the int8 kernel): int32 C, then the dequantisation pass
  if ((rc = h->qc.reserve((size_t)m * nb * sizeof(int32_t))) != MMH_OK) return rc;
  int32_t *qc = static_cast<int32_t *>(h->qc.p);
  if ((rc = launch_ab_modes(h, h->igemm_mode, m, n, k, qa, ka, qb, nb, qc, nb, 0, s)) < 0) return rc;
  if (rc == 1) HIP_TRY(launch_igemm_s8(m, n, k, qa, ka, qb, nb, qc, nb, 0, s, h->igemm_mode, cus));
  hipLaunchKernelGGL(dequantize_kernel, dim3(quant_rows_grid(m, 0)), dim3(256), 0, s, qc, m, n, nb,
                     scales, scales + 1, dC, ldc);
  HIP_TRY(hipGetLastError());
  return MMH_OK;
}

}  // extern "C"
