/*
 * mmult_hip.h -- C ABI of libmmult_hip.so, the MI355X (gfx950) backend behind
 * the reference's MY_MMult entry point.
 *
 * Everything here is `extern "C"`, plain pointers and sizes, int status codes
 * (0 = success, negative = MMH_ERR_*); nothing ever calls exit().  Matrices are
 * ROW-MAJOR fp32 with explicit leading dimensions (elements per row), exactly
 * as the reference harness passes them (cuda/test_MMult.cpp:62,102 --
 * lda=k, ldb=n, ldc=n).  Unlike the reference's CUDA kernels (which ignore
 * lda/ldb/ldc, e.g. cuda/MMult_cuda_3.cu:11,52) the leading dimensions ARE
 * honoured, and any m, n, k >= 0 is accepted (the reference kernels >= _3 have
 * no bounds checks and need multiples of the tile: cuda/MMult_cuda_9.cu:30-125).
 *
 * Which reference interface each entry point replaces (paths relative to
 * /root/reference):
 *
 *   mmh_sgemm         device-pointer MY_MMult, asynchronous on a stream:
 *                     cuda/test_MMult.cpp:13-14,102 and the launcher
 *                     cuda/MMult_cuda_12.cu:228-235 (C = A*B, overwrite);
 *                     accumulate=1 gives the host flavour's C = A*B + C
 *                     (armv7/MMult0.c:9-24, aarch64/MMult0.cpp:3-19).
 *   mmh_sgemm_host    host-pointer MY_MMult (armv7/test_MMult.c:8,76;
 *                     aarch64/test_MMult.cpp:17,113): does H2D, kernel, D2H.
 *   mmh_sgemm_host_timed  the third flavour of the symbol, `float MY_MMult(m, n, k, a, b, c)`
 *                     of the vulkan directory (vulkan/test_MMult.cpp:10,55): host pointers,
 *                     C = A*B, RETURNS the device time of the GEMM in milliseconds
 *                     (vulkan/MMult_vk_3.cpp:38-46 brackets the dispatch with timestamps).
 *   mmh_create/destroy the cublasHandle_t lifetime in the harness
 *                     (cuda/test_MMult.cpp:43-44,142).
 *   mmh_set_kernel    the makefile's `NEW := MMult_cuda_N` selection
 *                     (cuda/makefile:1-3,25), as a run-time switch.
 *   mmh_shard_rows / mmh_shard_create, _sgemm, _destroy / mmh_sgemm_sharded
 *                     no reference analogue (single device only,
 *                     cuda/test_MMult.cpp:24-25); BASELINE.json config 4.
 *   mmh_igemm_s8      no reference code (aarch64-int8/ is an empty submodule,
 *                     README.md:71-85); BASELINE.json config 5.
 *   mmh_sgemm_rocblas vendor comparator, cuda/MMult_cuBLAS_1.cpp:11-19 (cublasSgemm -> rocblas_sgemm).
 *   mmh_sgemm_hipblaslt  the second vendor comparator, cuda/MMult_cuBLAS_2.cpp:11-26 (cublasGemmEx with
 *                     fp32 compute -> hipblasLtMatmul, HIPBLAS_COMPUTE_32F).
 *   mmh_warm          what cublasCreate does for the reference before its timed loop
 *                     (cuda/test_MMult.cpp:43-44): every one-off a first launch would pay.
 *   mmh_probe_*       peak probes, the idea of aarch64/gflops_benchmark/main.c:19-25
 *                     and vulkan/benchmark/{gflops_fmla,gmem_bandwidth}.cpp.
 *
 * The C++-linkage forwarder `void MY_MMult(int,int,int,float*,int,float*,int,
 * float*,int)` (mangled _Z8MY_MMultiiiPfiS_iS_i, what aarch64/test_MMult.cpp:17
 * links against) lives in how-to-optimize-gemm_amd/harness/MMult_hip.cpp.
 */
#ifndef MMULT_HIP_H_
#define MMULT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmh_context *mmh_handle_t;

/* status codes */
#define MMH_OK 0
#define MMH_ERR_INVALID_ARG (-1) /* negative size, ld < row length, NULL pointer */
#define MMH_ERR_HIP (-2)         /* a HIP runtime call or kernel launch failed   */
#define MMH_ERR_NO_DEVICE (-3)   /* no gfx950 device visible                     */
#define MMH_ERR_UNSUPPORTED (-4) /* feature not built in (e.g. rocBLAS, RCCL)    */
#define MMH_ERR_ALLOC (-5)       /* device or host allocation failed             */
#define MMH_ERR_COMM (-6)        /* RCCL call failed                             */

/* kernel variants (the reference's "NEW := MMult_xxx" ladder, MI355X edition) */
#define MMH_KERNEL_AUTO 0        /* the K2W tiles (64x64 / 96x96 / 128x64 / 128x128) or the 256x256 tile, plain or
                                    stream-K: whichever a cost table fitted to measurements prices lowest for
                                    the shape (csrc/policy.hip, mmh_auto_plan) */
#define MMH_KERNEL_VALU 1        /* K1: LDS-tiled, VALU fma only: the 128x128 (8x8 outputs per thread), 128x64 or 64x64 (4x4)
                                    tile, whichever fills its last round of CUs best.  Whole-tile shapes run K1W (round 5:
                                    the staging done by loader waves' LDS-DMA, csrc/sgemm_valu_dma5.hpp), the others K1's
                                    guarded register-staged kernels; the same chain, the same bits */
#define MMH_KERNEL_VALU_128X128 13 /* K1 with the 128x128 tile always (the rung BASELINE config 2 names) */
#define MMH_KERNEL_VALU_64X64 14   /* K1 with the 64x64 tile always                                      */
#define MMH_KERNEL_VALU_128X64 9   /* K1W's 128x64 tile (8x4 outputs per thread; whole-tile shapes -- others run 128x128) */
#define MMH_KERNEL_MFMA 2        /* K2: 128x128 block tile on v_mfma_f32_16x16x4_f32; K-slice
                                    hand-over pipelined across the barrier, staging ops dealt
                                    out between MFMAs, buffer-descriptor loads; tile counts
                                    that do not divide the chip run as a persistent chained
                                    stream-K launch (K2p) -- same bits                      */
#define MMH_KERNEL_MFMA_256 3    /* K2b: 256x128 block tile, 8 waves                        */
#define MMH_KERNEL_NAIVE 4       /* one thread per C element (cuda/MMult_cuda_2.cu analogue) */
#define MMH_KERNEL_MFMA_SIMPLE 5 /* K2a: MFMA tile, plain double buffering                      */
#define MMH_KERNEL_MFMA_PIPE 6   /* K2b': + K-slice hand-over pipelined across the barrier, but
                                    compiler-scheduled staging and 64-bit global loads      */
#define MMH_KERNEL_MFMA_TILES 10 /* K2 always as one workgroup per tile (no stream-K), for A/B      */
#define MMH_KERNEL_MFMA_128X64 8 /* K2 with a 128x64 block tile, 4 waves of 64x32                   */
#define MMH_KERNEL_MFMA_256X256 12 /* K2 with a 256x256 block tile, 8 waves of 128x64 (1 WG/CU)       */
#define MMH_KERNEL_MFMA_64X64 11 /* K2 with a 64x64 block tile, 4 waves of 32x32, 128-deep K-slices  */
/* K2L (sgemm_dma.hpp): tiles fed entirely by LDS-DMA (buffer_load ... lds for both operands, a ring of
 * three 32-deep K-slice buffers = 48 / 72 / 96 KiB, i.e. 3 / 2 / 1 workgroups per CU, counted vmcnt waits,
 * no registers -> LDS stores at all) -- the register-staged
 * packing stage is what bounds the small tiles.  Same chain, same bits; any shape and 4-byte alignment (guarded
 * instantiations, round 3); what MMH_KERNEL_AUTO ran in round 3.  Since round 5 they are candidates of AUTO's cost table
 * beside K2W: their 256-thread workgroups and shorter preamble win on small shapes (m or n below ~600, K below ~500). */
#define MMH_KERNEL_MFMA_64X64_DMA 25
#define MMH_KERNEL_MFMA_128X64_DMA 27
#define MMH_KERNEL_MFMA_128X128_DMA 28
/* K2W (sgemm_dma5.hpp, round 4): K2L's tiles with LOADER waves that do nothing but the LDS-DMA (the pieces of a
 * K-slice, two slices ahead; two loaders for the 64x64 tile, four for the 128-wide and the 96x64 tiles, one for 96x96), so
 * that the four MFMA waves never
 * stall on a vector-memory issue; under stream-K the loaders walk the parts of a range as ONE stream of slices
 * (MMH_OPT_STREAMK_CHAIN).  Thin edge tiles skip the MFMAs of 16-row / 16-column blocks that hold no element.
 * Same chain, same bits. */
#define MMH_KERNEL_MFMA_64X64_DMA5 29
#define MMH_KERNEL_MFMA_128X64_DMA5 30
#define MMH_KERNEL_MFMA_128X128_DMA5 31
/* ... and a whole-round tile of 32 i x 32 j (wave tiles of 48 x 48; B fragments column-blocked): 96x96 lands
 * N = 1536 of the reference sweep (cuda/parameters.h:5-7) on exactly 256 tiles.  One workgroup per tile only (no
 * stream-K form).  (160x96 was built and measured too -- N = 1920 -- and loses to the chained stream-K launch of
 * the 128-wide tiles; it lives in the tools build, profiles/r04_notes.md.) */
#define MMH_KERNEL_MFMA_96X96_DMA5 7
#define MMH_KERNEL_MFMA_96X64_DMA5 26  /* round 5: 96x64 (wave tile 48x32), one workgroup per tile only: N = 1152 is 216 of them */
/* 160x160 (wave tiles of 80 x 80, four loaders, 120 KiB ring: one workgroup per CU; one workgroup per tile only).
 * Rounds 4-5 measured it BEHIND the 128x128 tile's stream-K launch (N = 2560: 140.8 against 145.2 TFLOP/s) and kept it in
 * the tools build: its ten single-float fragment reads per k-step left as a block in front of the 25 MFMAs, with a lone
 * wave per SIMD issuing nothing else meanwhile.  Since round 6 every K2W tile spreads its fragment reads behind the k-step's
 * first MFMAs (sgemm_dma5.hpp, RS): N = 2560 -- exactly 256 tiles -- 149.4, N = 5120 152.0 (profiles/r06_notes.md section 8). */
#define MMH_KERNEL_MFMA_160X160_DMA5 100
/* (Tools build only -- libmmult_hip_ab.so, never this library: K2M, the same LDS-DMA ring feeding
 * v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x1_2b_f32 (ids 48-51, 60-62; measured slower than the 16x16x4 tiles,
 * profiles/r04_notes.md), the one-loader and 160-wide forms of K2W (64, 68, 72, 79, 80), the scheduling A/Bs and the
 * timing-only ablations (16-19, 21-24, 32-47, 52-59).  mmh_set_kernel of the product library rejects them all.) */
/* OPT-IN split-K (sgemm_mfma.hpp K2s): the K range of every tile runs as S concurrent parts whose
 * partial tiles are summed in part order.  Deterministic, inside the reference harness's tolerance,
 * but NOT the one-chain-per-element bits every other variant returns; never chosen unless asked for
 * (these ids, or MMH_OPT_SPLITK with MMH_KERNEL_AUTO).  Whole-tile, 16-byte-aligned shapes only,
 * anything else runs the chain kernels. */
#define MMH_KERNEL_MFMA_SPLITK 15        /* 128x128 tiles, MMH_OPT_SPLITK parts (<= 1: as many as fill the chip) */
#define MMH_KERNEL_MFMA_SPLITK_128X64 20 /* the same on 128x64 tiles                                            */
/* The product library accepts exactly the ids above: every one of them returns correct results.
 * The scheduling A/B variants (ids 16-19) and the TIMING-ONLY ablation builds whose results are
 * wrong (21-24, 32-44; int8 modes 10-13), and the int8 rungs mode 0 never reaches (modes 1, 3, 4), exist only in libmmult_hip_ab.so, a tools-only build of the
 * same sources (-DMMH_AB_BUILD; how_to_optimize_gemm_amd.build.build_ab_library(), tools/ab_bench.py).
 * mmh_is_ab_build() tells the two apart.  See profiles/r01_ablation.md. */

/* Library / device ------------------------------------------------------- */
const char *mmh_strerror(int status);
/* Text of the last HIP/RCCL error seen on this thread ("" if none). */
const char *mmh_last_error(void);
/* Which kernel configuration the last mmh_sgemm on this thread launched (tile, wave tile,
 * grid, plain / stream-K / guarded) -- what MMH_KERNEL_AUTO chose. */
const char *mmh_last_launch(void);
int mmh_version(void);                 /* 100*major + minor */
int mmh_is_ab_build(void);             /* 0: the product library; 1: libmmult_hip_ab.so (tools only) */
int mmh_device_count(int *count);
/* name must hold >= 256 bytes; cu_count / clock_mhz may be NULL. */
int mmh_device_info(int device, char *name, int *cu_count, int *clock_mhz);

/* A handle owns device workspaces (host-flavour staging, stream-K partial-tile slots and hand-off words,
 * int8 / quantisation scratch) that consecutive calls reuse: use one handle per host thread.
 *
 * mmh_create also WARMS the handle (unless the environment says MMH_LAZY=1): every kernel MMH_KERNEL_AUTO can
 * pick is run once on one tile of scratch -- code objects loaded, > 64 KiB LDS opted into, residency queried --
 * and the stream-K workspaces are allocated, so that the first mmh_sgemm of a process is a launch and nothing
 * else (the reference creates its cuBLAS handle before its timed loop for the same reason,
 * cuda/test_MMult.cpp:43-44).  mmh_warm does the same for a lazily created handle; it is idempotent.
 *
 * Streams.  Calls on ONE stream through one handle are ordered by the stream (as cublasHandle_t with
 * cublasSetStream).  The stream-K / split-K workspaces exist once per stream the handle has launched on, so
 * launches on different streams share nothing, may overlap, and need no ordering between them; the handle
 * never touches a stream again once the call that launched on it has returned -- destroying a stream the
 * handle has used is fine.  (More than eight streams per handle: the least recently used stream's set is
 * released after a device-wide synchronisation.)  The other per-handle scratch (host-flavour staging, int8 /
 * quantisation buffers) is NOT per stream: use one handle per host thread, and one handle per stream for
 * mmh_sgemm_host / mmh_igemm_s8 / mmh_qgemm_f32 calls that should overlap.
 *
 * Progress guarantee.  Stream-K launches are persistent grids whose workgroups hand partial tiles to each
 * other, but a workgroup only ever waits for one that is ALREADY RUNNING and whose remaining work before the
 * hand-over depends on nobody (sgemm_mfma.hpp, K2p: a tile's word says whether the head's owner has started;
 * if it has not, the tail's owner marks the word and leaves, and the head's owner finishes the tile itself).
 * A launch therefore completes with any number of its workgroups resident, in any dispatch order: several
 * handles may run stream-K launches on concurrent streams, next to RCCL kernels or anything else that
 * occupies CUs -- co-residency changes the speed, never the result, and there is no time-out to report
 * (MMH_OPT_STREAMK_DELEGATIONS counts the hand-overs that took the slow path).  Only the OPT-IN split-K
 * kernels wait for workgroups that may not be resident -- bounded, raising the sticky error below.
 *
 * hipGraphs.  Launches capture into a graph once the CAPTURE STREAM owns a stream-K workspace set that is large
 * enough: nothing can be allocated while a stream is capturing, and a graph must not share hand-off words and
 * partial-tile slots with launches it is not ordered against, so a captured stream-K launch uses the capture stream's
 * own set or is refused (MMH_ERR_UNSUPPORTED) -- sets are never shared between streams.  A stream gets its set from
 * mmh_reserve_stream(handle, stream, m, n, k) (eager; sized for anything MMH_KERNEL_AUTO or a forced tile can launch
 * for that shape) or from one eager call of the shape on that stream; a shape that was never launched eagerly is
 * captured with plain-order ranges, otherwise the launch records the upload of its phase-order tables as a node of
 * the graph and pins them.  A set a graph points at is never released; a buffer of it that must grow is replaced and
 * the old one lives as long as the handle.  While a graph with stream-K launches RUNS it owns the capture stream's
 * set, like any buffer it was captured with: do not launch the same handle eagerly on the capture stream while a replay
 * may be running on another stream.  Synchronise a stream before destroying it (its set is found by the stream's
 * handle value, which the runtime may reuse).  Every entry point runs on the handle's device and restores the
 * caller's current device before it returns. */
int mmh_create(mmh_handle_t *handle, int device);
int mmh_destroy(mmh_handle_t handle);
int mmh_warm(mmh_handle_t handle);
int mmh_reserve_stream(mmh_handle_t handle, void *stream, int m, int n, int k);
int mmh_set_kernel(mmh_handle_t handle, int kernel);
int mmh_get_kernel(mmh_handle_t handle, int *kernel);
/* Name of a kernel variant ("MMult_hip_mfma", ...), NULL if unknown; and the id of a short name ("mfma",
 * "auto", "mfma_128x64_dma", ...: the name without its "MMult_hip_" prefix), -1 if unknown. */
const char *mmh_kernel_name(int kernel);
int mmh_kernel_id(const char *short_name);

/* Options.  MMH_OPT_STREAMK (default 1): let MMH_KERNEL_MFMA/AUTO run tile counts
 * that do not divide the chip as ONE persistent chained stream-K launch (bit-identical
 * results) -- from 128x128 tiles up whenever the count is ragged, for smaller tiles when the plain
 * launch would leave a round more than 7 % empty; 0 never, 2 whenever the count is ragged.
 * MMH_OPT_STREAMK_TIMEOUTS: the handle's STICKY error.  Stream-K launches cannot time out (their hand-over is
 * wait-free); the finisher of an OPT-IN split-K tile waits, bounded, for its partial tiles, and a wait that
 * times out (never seen outside fault injection) adds to a host-visible word and stops without storing; from
 * then on EVERY mmh_* call on the handle returns MMH_ERR_HIP -- the launch that timed out produced an invalid
 * result and nobody has to poll to learn it.  get: synchronises the device and returns the count; set 0:
 * synchronises and clears it. */
#define MMH_OPT_STREAMK 1
#define MMH_OPT_STREAMK_TIMEOUTS 2
/* MMH_OPT_IGEMM_MODE: 0 (default) B read in place (LDS-DMA of its row-major slices, fragments by
 * ds_read_b64_tr_b8) for 4-byte aligned operands -- 256x256 tiles on the ping-pong kernel K3p (igemm_s8_pp.hpp) where
 * their rounds are cheaper, 128x128 tiles (K3t) otherwise; operands that are not 4-byte aligned are first copied
 * into dense workspace images; 2 the correctness-first kernel (also what anything beyond the descriptors' 2 GiB
 * window runs); 5 / 6 the lockstep in-place kernel K3t with 128x128 / 256x256 tiles, 8 / 9 K3p as a persistent launch /
 * one workgroup per tile (mode 0 picks by K), 7 K3p on v_mfma_i32_16x16x32_i8 -- the instruction BASELINE.json
 * configs[4] names: the same integers at half the matrix pipe's rate (A/B switches).
 * (Modes 1, 3, 4 -- the in-kernel-transpose and packed-B rungs of rounds 1-2 -- and 10..13, timing-only ablations with
 * wrong results, exist in libmmult_hip_ab.so only; the product library rejects them.) */
#define MMH_OPT_IGEMM_MODE 3
/* MMH_OPT_SPLITK (default 0 = off): 1 lets MMH_KERNEL_AUTO run shapes with fewer 128x128 tiles than
 * CUs as a split-K launch with as many concurrent K parts as fill the chip; 2..16 asks for that many
 * parts.  See MMH_KERNEL_MFMA_SPLITK: deterministic, within the harness tolerance, not the chain's bits. */
#define MMH_OPT_SPLITK 4
/* MMH_OPT_HOST_PANELS (default -1 = automatic): row panels of the host-pointer flavour's copy/compute
 * pipeline (mmh_sgemm_host); 0 or 1 = the plain copy-in, GEMM, copy-out sequence; 2..16 panels. */
#define MMH_OPT_HOST_PANELS 5
/* Test hooks.  MMH_OPT_STREAMK_SPIN_LIMIT: bound of the split-K finisher's wait in units of 1024 polls (default
 * 65536, seconds).  MMH_OPT_FAULT_INJECT (default 0): 1 = split-K producers do not announce their
 * partial tiles, so that every finisher times out and the sticky error path can be exercised.
 * (Diagnostic environment switches, read at mmh_create: MMH_NO_PIN=1 -- persistent launches then do not
 * ask for 160 KiB / w of LDS to pin w workgroups per CU; MMH_I8_GRID_CAP=g -- the int8 ping-pong kernel's
 * persistent grid is g workgroups, so that a test's small shapes walk several tiles per workgroup.) */
#define MMH_OPT_STREAMK_SPIN_LIMIT 6
#define MMH_OPT_FAULT_INJECT 7
/* MMH_OPT_STREAMK_ORDER (default 1): stream-K launches with >= 1.8 tiles per workgroup take their ranges
 * in K-PHASE order and their tiles in a matching placement (two small per-shape tables, built on the host
 * at the shape's first eager launch -- one synchronising copy -- and cached in the handle; launches captured
 * into a hipGraph run with ranges in plain order), so that
 * workgroups that are neighbours on the chip walk K in step on neighbouring tiles and share operand slices
 * in L2 as a plain launch does (hit rate 22-35 % -> 75 %).  Same chain, same bits; 0 = ranges in plain
 * order (the A/B baseline; environment MMH_NO_SK_ORDER=1 does the same at mmh_create). */
#define MMH_OPT_STREAMK_ORDER 8
/* MMH_OPT_DMA_EDGE (default 2): which shapes MMH_KERNEL_AUTO and the *_DMA ids may run on the GUARDED LDS-DMA
 * tiles (sgemm_dma.hpp): 0 none (ragged or unaligned shapes run the register-staged tiles, the round-2
 * behaviour), 1 any m, n, k whose A and B rows are 16-byte aligned, 2 any 4-byte aligned operands. */
#define MMH_OPT_DMA_EDGE 9
/* MMH_OPT_STREAMK_DELEGATIONS (diagnostic, read-only; set 0 resets): how many stream-K hand-overs on this handle were
 * finished by the HEAD's owner because the tail's owner got there first and left (sgemm_mfma.hpp, K2p).  0 while the
 * persistent grids are co-resident; > 0 says a launch ran with part of its grid queued -- correct, only slower.
 * get synchronises the device. */
#define MMH_OPT_STREAMK_DELEGATIONS 10
/* MMH_OPT_RIM (default 0 = off, at most 16; an experiment that is kept for its measurements, profiles/r03_notes.md
 * section 6): MMH_KERNEL_AUTO runs a shape whose m and n are at most this many rows / columns past a multiple of 64
 * (N = 1025) as the tiles of the TRIMMED shape plus "the rim" -- the thin strips of C beyond it, computed on the vector
 * ALU by extra workgroups of the same launch (one fused-multiply-add chain over ascending k per element: the same bits
 * as the tiles) -- instead of paying a whole extra row and column of edge tiles.  Applies where the trimmed shape is a
 * plain launch of the 64-wide LDS-DMA tiles (sgemm_dma.hpp) and tiles and rim units are all resident at once.  Off by
 * default because it does not pay: alone the tiles of 1024 x 1024 x 1025 take 20 us and the rim 13 us, together 28-29 us
 * -- against 27 us for the plain launch of 17 x 17 edge tiles. */
#define MMH_OPT_RIM 11
/* MMH_OPT_STREAMK_CHAIN (default 1): stream-K launches of the K2W tiles (sgemm_dma5.hpp; launch_dma5.hip picks the
 * instantiation) run the parts of a workgroup's range as ONE stream of K-slices -- a part's last slices fetch the next part's first ones -- instead of starting every
 * part with an empty pipeline.  Same bits; 0 = the unchained form (the A/B baseline). */
#define MMH_OPT_STREAMK_CHAIN 12
/* MMH_OPT_PERSIST (default 0): 1 = tile counts that ARE whole rounds of the persistent grid (>= 2 tiles per
 * workgroup) also run as the persistent launch: every workgroup walks its tiles in the phase tables' level order,
 * so the co-resident workgroups of an XCD start together and stay in K lock-step (what a fresh workgroup per tile
 * loses to dispatch stagger), and the K2W loaders fetch the next tile's first slices under the current tile's
 * store.  No partial tiles are handed over in such a launch.  Applies to FORCED kernels only: under MMH_KERNEL_AUTO the
 * cost table has decided the launch form (it prices no persistent whole-round launch: measured 0.3 % slower). */
#define MMH_OPT_PERSIST 13
/* MMH_OPT_RIM5 (tools build only; the product accepts 0): shapes ONE row and / or column past a multiple of 64 run the
 * 64x64 K2W tiles of the TRIMMED shape and an extra wave per edge tile computes the rim on the vector ALU out of the
 * K-slices in LDS.  Correct to the bit, and measured 2.2x slower per edge tile than a whole tile (the f32 MFMA shares
 * the vector ALU's FMA lanes): the product runs thin edge tiles instead (sgemm_dma5.hpp). */
#define MMH_OPT_RIM5 14
int mmh_set_option(mmh_handle_t handle, int option, int value);
/* The two tables of a phase-ordered stream-K launch (MMH_OPT_STREAMK_ORDER) for `tiles` tile slots of `nk`
 * K-slices on `grid` persistent workgroups, computed on the host (no device needed): order[grid] = the range
 * each chip position takes, place[tiles] = the tile computed in each slot.  For tests and tools. */
int mmh_streamk_plan(long tiles, int nk, int grid, int *order, int *place);
/* What MMH_KERNEL_AUTO would run for a shape -- the reference's `NEW := MMult_xxx` makefile choice (cuda/makefile:1-3)
 * made per call -- computed on the host by the launch path's own functions (no device needed, nothing launched):
 * *kernel = the MMH_KERNEL_* id of the tile, *tiles = how many of them the shape takes (edge tiles included),
 * *streamk_grid = the persistent workgroups of a stream-K launch, 0 for one workgroup per tile, -1 where the answer
 * needs the device's occupancy query (the register-staged 128x128 / 128x64 / 64x64 tiles).  base_align: 16 or 4, the
 * alignment of the operands' base addresses; cu_count <= 0: 256.  Handle options at their defaults.  For tests,
 * tools and callers that want to know before they launch. */
int mmh_auto_plan(int m, int n, int k, int lda, int ldb, int ldc, int base_align, int cu_count, int *kernel, long *tiles,
                  int *streamk_grid);
int mmh_get_option(mmh_handle_t handle, int option, int *value);

/* The hot path ------------------------------------------------------------ */
/*
 * C[m x n] = A[m x k] * B[k x n]            (accumulate == 0)
 * C[m x n] = A[m x k] * B[k x n] + C        (accumulate != 0; C's value is the
 *                                            first term of each element's chain)
 * dA, dB, dC: device pointers on the handle's device.  Enqueued on `stream`
 * (a hipStream_t passed as void*, NULL = the null stream) and returns without
 * synchronising, like the reference launcher.  Each output element is an
 * fp32 fused-multiply-add chain over ascending k -- for every kernel variant.
 */
int mmh_sgemm(mmh_handle_t handle, int m, int n, int k, const float *dA, int lda,
              const float *dB, int ldb, float *dC, int ldc, int accumulate,
              void *stream);

/* Host-pointer flavour: stages A, B (and C when accumulating) to the device,
 * runs mmh_sgemm, copies C back, synchronises.  Staging buffers are cached in
 * the handle and grow on demand.  Large problems run as a row-panel pipeline (copy-in of panel
 * i+1, GEMM of panel i and copy-out of panel i-1 overlap on three streams; MMH_OPT_HOST_PANELS);
 * row panels are independent, so the bits are those of the single launch. */
int mmh_sgemm_host(mmh_handle_t handle, int m, int n, int k, const float *A, int lda,
                   const float *B, int ldb, float *C, int ldc, int accumulate);

/* The same call in the plain staged form (copy in, ONE launch, copy out) with the launch bracketed by
 * two events: *kernel_ms receives the device time of the GEMM alone, copies excluded -- what the
 * reference's Vulkan flavour of MY_MMult returns (vulkan/test_MMult.cpp:55 sums it over NREPEATS). */
int mmh_sgemm_host_timed(mmh_handle_t handle, int m, int n, int k, const float *A, int lda,
                         const float *B, int ldb, float *C, int ldc, int accumulate, float *kernel_ms);

/* int8 x int8 -> int32, C = A*B (+ C), row-major, inputs expected in
 * [-127,127]; bit-exact integer arithmetic on v_mfma_i32_16x16x64_i8.  4-byte aligned operands
 * (bases and leading dimensions) are read in place; others are first copied into dense aligned
 * images in a handle-owned workspace (or, for B only, packed transposed) -- see MMH_OPT_IGEMM_MODE. */
int mmh_igemm_s8(mmh_handle_t handle, int m, int n, int k, const int8_t *dA, int lda,
                 const int8_t *dB, int ldb, int32_t *dC, int ldc, int accumulate,
                 void *stream);

/* The callers either side of the int8 GEMM (chgemm's contract as the
 * reference README.md:71-85 words it; parity unpinned -- no reference code):
 *   mmh_quantize_sym_s8: q = clamp(rint(x * s), -127, 127), s = 127 / max|x| written to
 *                        *d_scale (device float); rows x cols windows with leading dims.
 *   mmh_qgemm_f32      : C_f32 = dequant( quant(A) * quant(B) ), per-tensor symmetric
 *                        scales, int32 accumulation, C = acc * (1 / (sa * sb)). */
int mmh_quantize_sym_s8(mmh_handle_t handle, int rows, int cols, const float *dX, int ldx,
                        int8_t *dQ, int ldq, float *d_scale, void *stream);
int mmh_qgemm_f32(mmh_handle_t handle, int m, int n, int k, const float *dA, int lda,
                  const float *dB, int ldb, float *dC, int ldc, void *stream);

/* Vendor comparator (rocBLAS sgemm, row-major via the swapped-operand trick
 * of cuda/MMult_cuBLAS_1.cpp:17-18).  MMH_ERR_UNSUPPORTED if librocblas
 * cannot be loaded. */
int mmh_sgemm_rocblas(mmh_handle_t handle, int m, int n, int k, const float *dA, int lda,
                      const float *dB, int ldb, float *dC, int ldc, void *stream);
/* The second vendor comparator (hipBLASLt, fp32 compute, same swapped-operand trick; the algorithm the
 * library's own heuristic ranks first, looked up once per shape and cached in the handle; 64 MiB of
 * workspace offered).  MMH_ERR_UNSUPPORTED if libhipblaslt cannot be loaded or offers no fp32 algorithm. */
int mmh_sgemm_hipblaslt(mmh_handle_t handle, int m, int n, int k, const float *dA, int lda,
                        const float *dB, int ldb, float *dC, int ldc, void *stream);

/* Multi-GPU row-panel shard ---------------------------------------------- */
/* Rows [*row0, *row0 + *rows) of C (and A) owned by `rank` of `nranks`:
 * contiguous panels, multiples of 128 rows while whole tiles remain, the
 * remainder spread from rank 0 up.  Pure host arithmetic. */
int mmh_shard_rows(int m, int nranks, int rank, int *row0, int *rows);

/* Single-process form with a HANDLE (one RCCL communicator, one stream, one product handle and one
 * set of device buffers per device, all created once): host A, B, C; A row panels go to their owning
 * devices (one host thread per device: every device has its own PCIe link), B goes to the first
 * device and is replicated with ONE ncclBroadcast over xGMI, the row-panel GEMMs run concurrently
 * (stream-K and tile choice per device, as mmh_sgemm), C panels are copied back.
 *   devices: `ngpus` distinct device ordinals, NULL = 0..ngpus-1.  MMH_ERR_NO_DEVICE when fewer are
 *            visible (never a silent fallback to fewer), MMH_ERR_UNSUPPORTED without librccl (ngpus > 1).
 *   mmh_shard_sgemm: gemm_reps >= 1 back-to-back GEMM launches per device (the reference harness's
 *            NREPEATS loop); timings_ms (may be NULL) receives {h2d, bcast, gemm per rep, d2h}.
 *   mmh_shard_sgemm_streamed (round 5): the same with B travelling in `b_chunks` runs of K (whole 128-deep blocks; 0 or
 *            1 = ONE broadcast, at most 64) on a second, higher-priority stream per device while the chunks already
 *            landed are consumed -- C = A[:, chunk] B[chunk, :] + C with C's value as the first term of each chain, i.e.
 *            the unchunked launch's bits.  With gemm_reps == 1 that pass IS the launch: the C copied back is the chunked
 *            pass's and timings_ms[2] its time; gemm_reps > 1 adds that many full-K launches behind it (gemm_reps + 1 in
 *            all) and reports their mean.  An error return drains the handle's streams first.
 *            timings_ms (may be NULL) receives EIGHT floats, every device phase taken with events on that device's own
 *            streams, the slowest device reported: {h2d (host clock), broadcast (first chunk out .. last chunk landed),
 *            gemm per full-K launch, d2h (host clock), OVERLAPPED (broadcast start .. end of the first GEMM pass: what
 *            a caller who has to pay for B waits), chunks used, the first GEMM pass alone, host clock around the
 *            device phases}.  mmh_shard_sgemm is this with b_chunks = 1 and the first four figures.
 *   mmh_shard_info: rccl_ranks = ranks of the communicator (0 for ngpus == 1, where RCCL is not used). */
typedef struct mmh_shard *mmh_shard_t;
int mmh_shard_create(mmh_shard_t *shard, int ngpus, const int *devices);
int mmh_shard_destroy(mmh_shard_t shard);
int mmh_shard_set_kernel(mmh_shard_t shard, int kernel);
int mmh_shard_info(mmh_shard_t shard, int *ngpus, int *rccl_ranks);
int mmh_shard_sgemm(mmh_shard_t shard, int m, int n, int k, const float *A, int lda, const float *B,
                    int ldb, float *C, int ldc, int gemm_reps, float *timings_ms);
int mmh_shard_sgemm_streamed(mmh_shard_t shard, int m, int n, int k, const float *A, int lda, const float *B,
                             int ldb, float *C, int ldc, int gemm_reps, int b_chunks, float *timings_ms);
/* The K boundaries of the streamed broadcast as host arithmetic (no device): returns the number of chunks c actually
 * used (1 <= c <= min(b_chunks, ceil(k / 128), 64)) and fills k0[0 .. c] (k0 holds at least 65 ints): chunk i is rows
 * k0[i] .. k0[i + 1] - 1 of B, every interior boundary a multiple of 128. */
int mmh_shard_chunks(int k, int b_chunks, int *k0);
/* Page-lock (hipHostRegister) a host range that is about to be passed to mmh_shard_sgemm more than once -- A, B
 * and C of one sweep size: copies from / to pageable memory run at a fraction of the PCIe rate.  Unpin before
 * the memory is freed; mmh_shard_destroy unpins whatever is left.
 * (Test mode: with MMH_SHARD_SHARE_DEVICE=1 in the environment, a device list that names ONE device ngpus times
 * creates ngpus LOGICAL ranks on it -- B replicated by device copies, no RCCL -- so that the phase plumbing of
 * an N-rank shard, empty row panels included, runs on a box with one GPU.  Never entered implicitly.
 * Second test switch: with MMH_SHARD_FORCE_RCCL=1 a ONE-device shard (ngpus == 1) builds a one-rank communicator
 * (ncclCommInitAll) and mmh_shard_sgemm issues its ncclBroadcast on it -- rccl_ranks = 1 -- so that the RCCL branch
 * has run once before the first multi-GPU node sees it.) */
int mmh_shard_pin(mmh_shard_t shard, void *host, size_t bytes);
int mmh_shard_unpin(mmh_shard_t shard, void *host);
/* RCCL as this library sees it: loads librccl (dlopen) and returns ncclGetVersion's code in *version;
 * MMH_ERR_UNSUPPORTED when the library or one of the entry points the shard needs is missing.
 * Needs no GPU. */
int mmh_rccl_version(int *version);

/* One-shot convenience form of the above: creates a shard handle, runs once, destroys it (so it
 * pays the communicator's creation on every call -- use the handle form in a loop). */
int mmh_sgemm_sharded(int ngpus, int m, int n, int k, const float *A, int lda,
                      const float *B, int ldb, float *C, int ldc, int kernel,
                      float *timings_ms);

/* Measurement helpers ------------------------------------------------------ */
/* Mean milliseconds per call over `reps` back-to-back mmh_sgemm launches
 * bracketed by one hipEvent pair on `stream` (the reference's timing
 * convention, cuda/test_MMult.cpp:98-114), after `warmup` untimed calls. */
int mmh_time_sgemm(mmh_handle_t handle, int m, int n, int k, const float *dA, int lda,
                   const float *dB, int ldb, float *dC, int ldc, int warmup, int reps,
                   void *stream, float *ms_per_call);

/* The same measurement for a vendor comparator (the calls are issued from C, like mmh_time_sgemm's, so that
 * a 20 us kernel is not timed through an interpreter's call overhead). */
#define MMH_COMPARATOR_ROCBLAS 1
#define MMH_COMPARATOR_HIPBLASLT 2
int mmh_time_comparator(mmh_handle_t handle, int which, int m, int n, int k, const float *dA, int lda,
                        const float *dB, int ldb, float *dC, int ldc, int warmup, int reps, void *stream,
                        float *ms_per_call);

/* Per-launch milliseconds of `count` (<= 4096) back-to-back mmh_sgemm launches, one hipEvent pair
 * each: the clock-ramp trace (profiles/r02_clock_ramp.csv). */
int mmh_trace_sgemm(mmh_handle_t handle, int m, int n, int k, const float *dA, int lda,
                    const float *dB, int ldb, float *dC, int ldc, int count, void *stream,
                    float *ms_each);

/* Peak probes: sustained fp32 MFMA TFLOP/s (v_mfma_f32_16x16x4_f32 only, no
 * memory traffic), HBM copy GB/s (float4 stream copy, four loads in flight per thread,
 * non-temporal; read+write bytes) and HBM read GB/s (read-only twin, eight loads in flight). */
int mmh_probe_mfma_f32(mmh_handle_t handle, float *tflops);
int mmh_probe_mfma_i8(mmh_handle_t handle, float *tops);   /* v_mfma_i32_16x16x64_i8 only */
/* The vector ALU's fp32 FMA rate: v_pk_fma_f32 (packed != 0) or v_fma_f32 only, 64 independent accumulators per lane,
 * waves_per_simd (1..4) waves on every SIMD, no memory traffic -- the measured denominator of the K1 / K1W rung. */
int mmh_probe_valu_f32(mmh_handle_t handle, int packed, int waves_per_simd, float *tflops);
/* The same loop run back to back for at least `min_ms` (0..2000), reporting the last 2.3 ms launch:
 * random_operands != 0 gives every MFMA different pseudo-random inputs (the rate the power
 * manager sustains on real data), 0 keeps the constant operands of mmh_probe_mfma_i8. */
int mmh_probe_mfma_i8_sustained(mmh_handle_t handle, int random_operands, float min_ms, float *tops);
int mmh_probe_hbm_copy(mmh_handle_t handle, size_t bytes, float *gbps);
int mmh_probe_hbm_read(mmh_handle_t handle, size_t bytes, float *gbps);
/* LDS fragment reads and nothing else (the idea of vulkan/benchmark/smem_bandwidth.cpp:30-42), summed
 * over the chip in GB/s: width = bytes per lane, 16 (ds_read_b128), 8 (ds_read_b64), 4 (ds_read_b32)
 * or -8 (ds_read_b64_tr_b8, the transposing read of the in-place int8 kernel).  The roof for the 8- and
 * 16-byte reads is 256 B/clk/CU (MI355X_MICROARCH.md, LDS) = 157 TB/s at 2.4 GHz x 256 CUs, half that
 * for ds_read_b32. */
int mmh_probe_lds_read(mmh_handle_t handle, int width, float *gbps);

#ifdef __cplusplus
}
#endif
#endif /* MMULT_HIP_H_ */
