// test_MMult.cpp -- the sweep driver, MI355X edition.
//
// Control flow, timing convention, tolerance and stdout format are those of the
// reference driver (cuda/test_MMult.cpp:21-146; host flavour:
// armv7/test_MMult.c:16-103, aarch64/test_MMult.cpp:24-144):
//
//   for p = PFIRST..PLAST step PINC:            cuda/parameters.h:5-7
//     m,n,k := p unless fixed; lda=k ldb=n ldc=n cuda/test_MMult.cpp:56-62
//     a,b,cold <- random_matrix x3, cold=cref=0  :77-81 (same drand48 call pattern)
//     d_A,d_B <- H2D (untimed)                   :84-89
//     cref <- REF_MMult                          :94
//     event pair around NREPEATS x MY_MMult      :98-112   GFLOPS = 2mnk/t_mean :116-118
//     cold <- D2H; diff = compare_matrices       :121-123  |diff| > 0.5 -> exit(-1) :124-127
//     printf("%d %.2f %le \n", p, gflops, diff)  :128
//
// so an `output_<kernel>.m` written by `make run` is read by the reference's
// cuda/plot.py:5-28 unchanged.  Everything tunable is a run-time option
// (environment variable NAME or --NAME=value), defaults from parameters.h:
//
//   PFIRST PLAST PINC M N K NREPEATS LDA LDB LDC   sweep shape
//   KERNEL=auto|mfma|mfma256|mfma_256x256|mfma_128x64|mfma_64x64|mfma_64x64_dma|mfma_128x64_dma|mfma_128x128_dma|
//          mfma_pipe|mfma_simple|valu|valu_128x128|valu_64x64|naive|mfma_splitk|mfma_splitk_128x64 (the last two:
//          opt-in split-K) -- any short name mmh_kernel_id knows -- or rocblas | hipblaslt, the vendor comparators
//          (cuda/MMult_cuBLAS_1.cpp, cuda/MMult_cuBLAS_2.cpp)
//   SPLITK=<n>                 MMH_OPT_SPLITK for KERNEL=auto (0 off, 1 auto, 2..16 parts)
//   FLAVOUR=device|host|cpu|sharded
//                                device: C=A*B on device pointers (cuda/ flavour)
//                                host  : MY_MMult(m,n,k,a,lda,...) on host pointers, C+=A*B,
//                                        best-of-NREPEATS with dclock (armv7/aarch64 flavour)
//                                cpu   : MY_MMult := the serial triple loop, no GPU at all
//                                        (BASELINE.json config 1, plumbing check: diff = 0)
//                                sharded: C row panels over NGPUS devices of this process (ONE shard
//                                        handle for the whole sweep: communicator, streams and
//                                        buffers persist), B by one ncclBroadcast (BASELINE.json
//                                        config 4); GFLOPS from the GEMM phase (NREPEATS launches
//                                        per device), EXTENDED adds h2d/bcast/gemm/d2h ms and -- with
//                                        B_CHUNKS=<c> > 1: B travels in c K-chunks under the GEMMs that
//                                        consume them, mmh_shard_sgemm_streamed -- the overlapped
//                                        broadcast + first-GEMM-pass ms and the chunk count
//   INPUT=drand48|seed:<n>|mod3|mod2|ones           (cuda/random_matrix.cpp:9-15 variants)
//   REF=threads|serial|blas|skip  how cref is produced: the triple loop split over host threads
//                                (default), the literal serial loop, the host BLAS's cblas_sgemm
//                                (the cuda directory's oracle, cuda/REF_MMult.cpp:9-13), or not at
//                                all (diff column is -1)
//   WARMUP=<n>                 untimed launches before the timed loop (reference: 0)
//   WARMUP_MS=<ms>             keep launching untimed until this many milliseconds have passed (the
//                              sustained-clock form: between two sizes the driver spends seconds on
//                              the host -- inputs, REF -- and the GPU falls back to its idle clock;
//                              30 launches of a 0.1 ms kernel do not bring it back, 50 ms usually do);
//                              followed by untimed bursts of NREPEATS launches until two in a row agree
//                              within 1 % (at most 12): the rate has settled
//   TRIALS=<n>                 device flavour: time the NREPEATS launches n times and report the MEDIAN of
//                              the n means (default 1 = the reference's single measurement); the sustained
//                              sweeps under profiles/ use 3, so that one transient (another process's
//                              burst, a clock dip) does not own a point
//   EXTENDED=1                 extra columns: pct_of_fp32_mfma_peak ref_gflops ref_cores
//   JSON=<path>                besides the reference-format rows on stdout, one JSON object per size in
//                              <path> (a JSON array): p, m, n, k, gflops, diff, pct_of_fp32_mfma_peak,
//                              seconds, flavour, kernel and what the library actually launched
//                              (mmh_last_launch) -- the machine-readable side of the result file
//   PROBES=1                   before the sweep, measure the denominators on this device and print them on
//                              STDERR (stdout keeps the reference's format): MFMA-only fp32 TFLOP/s, HBM
//                              copy / read GB/s, LDS fragment-read GB/s -- the idea of
//                              aarch64/gflops_benchmark/main.c:19-25 and vulkan/benchmark/*.cpp
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "helper.h"
#include "parameters.h"
#include "utils.h"

void MY_MMult(int, int, int, float *, int, float *, int, float *, int);
void MY_MMult(mmh_handle_t, int, int, int, float *, int, float *, int, float *, int);

namespace {

struct Options {
  int pfirst = sweep_defaults::kFirstSize, plast = sweep_defaults::kLastSize, pinc = sweep_defaults::kStep;
  int m = sweep_defaults::kM, n = sweep_defaults::kN, k = sweep_defaults::kK;
  int nrepeats = sweep_defaults::kRepeats;
  int lda = sweep_defaults::kLda, ldb = sweep_defaults::kLdb, ldc = sweep_defaults::kLdc;
  int warmup = 0, warmup_ms = 0, extended = 0, ngpus = 1, splitk = 0, trials = 1, probes = 0, b_chunks = 1;
  std::string json;
  std::string kernel = "auto", flavour = "device", input = "drand48", ref = "threads";
};

const char *lookup(int argc, char **argv, const char *name) {
  const std::string key = std::string("--") + name + "=";
  for (int i = 1; i < argc; ++i)
    if (!std::strncmp(argv[i], key.c_str(), key.size())) return argv[i] + key.size();
  return std::getenv(name);
}
void opt_int(int argc, char **argv, const char *name, int &dst) {
  if (const char *v = lookup(argc, argv, name)) dst = std::atoi(v);
}
void opt_str(int argc, char **argv, const char *name, std::string &dst) {
  if (const char *v = lookup(argc, argv, name)) dst = v;
}

// KERNEL=<short name>: the library's own table (mmh_kernel_id), plus the two vendor comparators the
// reference links as MMult_cuBLAS_1 / MMult_cuBLAS_2 (cuda/makefile:1)
constexpr int kRocblas = -100, kHipblaslt = -101;
int kernel_id(const std::string &s) {
  if (s == "rocblas") return kRocblas;
  if (s == "hipblaslt") return kHipblaslt;
  const int id = mmh_kernel_id(s.c_str());
  if (id >= 0) return id;
  std::fprintf(stderr, "unknown KERNEL=%s\n", s.c_str());
  std::exit(EXIT_FAILURE);
}

void fill(const Options &o, int rows, int cols, float *buf, int ld) {
  if (o.input == "mod3") pattern_matrix(rows, cols, buf, ld, 3);
  else if (o.input == "mod2") pattern_matrix(rows, cols, buf, ld, 2);
  else if (o.input == "ones") pattern_matrix(rows, cols, buf, ld, 0);
  else random_matrix(rows, cols, buf, ld);
}

constexpr double kPeakTflops = 157.3;  // MI355X fp32 MFMA: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz

}  // namespace

int main(int argc, char **argv) {
  Options o;
  opt_int(argc, argv, "PFIRST", o.pfirst);  opt_int(argc, argv, "PLAST", o.plast);
  opt_int(argc, argv, "PINC", o.pinc);      opt_int(argc, argv, "M", o.m);
  opt_int(argc, argv, "N", o.n);            opt_int(argc, argv, "K", o.k);
  opt_int(argc, argv, "NREPEATS", o.nrepeats);
  opt_int(argc, argv, "LDA", o.lda);        opt_int(argc, argv, "LDB", o.ldb);
  opt_int(argc, argv, "LDC", o.ldc);        opt_int(argc, argv, "WARMUP", o.warmup);
  opt_int(argc, argv, "EXTENDED", o.extended);
  opt_int(argc, argv, "WARMUP_MS", o.warmup_ms);
  opt_int(argc, argv, "TRIALS", o.trials);
  opt_int(argc, argv, "PROBES", o.probes);
  opt_str(argc, argv, "JSON", o.json);
  if (o.trials < 1) o.trials = 1;
  opt_int(argc, argv, "NGPUS", o.ngpus);
  opt_int(argc, argv, "B_CHUNKS", o.b_chunks);
  opt_int(argc, argv, "SPLITK", o.splitk);
  opt_str(argc, argv, "KERNEL", o.kernel);  opt_str(argc, argv, "FLAVOUR", o.flavour);
  opt_str(argc, argv, "INPUT", o.input);    opt_str(argc, argv, "REF", o.ref);
  if (o.pinc <= 0 || o.nrepeats <= 0) { std::fprintf(stderr, "bad PINC/NREPEATS\n"); return 2; }
  if (!o.input.compare(0, 5, "seed:")) srand48(std::atol(o.input.c_str() + 5));

  const bool cpu_only = o.flavour == "cpu";
  const bool host_flavour = o.flavour == "host";
  const bool sharded = o.flavour == "sharded";
  mmh_handle_t handle = nullptr;
  mmh_shard_t shard = nullptr;
  hipEvent_t start{}, stop{};
  const int kid = kernel_id(o.kernel);
  if (!cpu_only) {
    char name[256];
    int cus = 0, mhz = 0;
    MMH_CHECK(mmh_create(&handle, 0));
    MMH_CHECK(mmh_device_info(0, name, &cus, &mhz));
    std::printf("GPU Device %d: \"%s\" with %d CUs @ %d MHz\n\n", 0, name, cus, mhz);
    if (o.probes) {
      float mfma = 0, copy = 0, read = 0, lds = 0;
      MMH_CHECK(mmh_probe_mfma_f32(handle, &mfma));
      MMH_CHECK(mmh_probe_hbm_copy(handle, (size_t)1 << 30, &copy));
      MMH_CHECK(mmh_probe_hbm_read(handle, (size_t)1 << 30, &read));
      MMH_CHECK(mmh_probe_lds_read(handle, 16, &lds));
      std::fprintf(stderr, "probes: mfma_f32 %.1f TFLOP/s, hbm copy %.0f GB/s, hbm read %.0f GB/s, lds read %.0f GB/s"
                           " (%.1f B/clk/CU at %d MHz)\n",
                   mfma, copy, read, lds, lds * 1e9 / ((double)mhz * 1e6 * cus), mhz);
    }
    if (kid >= 0) MMH_CHECK(mmh_set_kernel(handle, kid));
    if (o.splitk) MMH_CHECK(mmh_set_option(handle, MMH_OPT_SPLITK, o.splitk));
    if (sharded) {
      // fails (MMH_ERR_NO_DEVICE) when fewer than NGPUS devices are visible: never fewer, silently
      MMH_CHECK(mmh_shard_create(&shard, o.ngpus, nullptr));
      MMH_CHECK(mmh_shard_set_kernel(shard, kid >= 0 ? kid : MMH_KERNEL_AUTO));
    }
    if (host_flavour) setenv("MMULT_KERNEL", o.kernel.c_str(), 1);
    HIP_CHECK(hipEventCreate(&start));
    HIP_CHECK(hipEventCreate(&stop));
  } else {
    std::printf("CPU only: MY_MMult = serial triple loop (plumbing)\n\n");
  }
  std::printf("MY_MMult = [\n");

  std::FILE *json_out = nullptr;
  int json_rows = 0;
  if (!o.json.empty()) {
    json_out = std::fopen(o.json.c_str(), "w");
    if (!json_out) {
      std::fprintf(stderr, "JSON=%s: cannot open for writing\n", o.json.c_str());
      return EXIT_FAILURE;
    }
    std::fprintf(json_out, "[");
  }
  for (int p = o.pfirst; p <= o.plast; p += o.pinc) {
    const int m = o.m == -1 ? p : o.m, n = o.n == -1 ? p : o.n, k = o.k == -1 ? p : o.k;
    const int lda = o.lda == -1 ? k : o.lda, ldb = o.ldb == -1 ? n : o.ldb,
              ldc = o.ldc == -1 ? n : o.ldc;
    if (lda < k || ldb < n || ldc < n) { std::fprintf(stderr, "leading dimension too small\n"); return 2; }
    const double flops = 2.0 * m * n * (double)k;

    // Host buffers.  The generator writes column-major into a dense (rows x cols)
    // block exactly as the reference calls it (lda = m / k / n); the block is then
    // read row-major.  With padded leading dimensions it is embedded row by row.
    std::vector<float> dense_a((size_t)m * k), dense_b((size_t)k * n),
        burn((size_t)std::max(m, n) * n);
    fill(o, m, k, dense_a.data(), m);
    fill(o, k, n, dense_b.data(), k);
    fill(o, m, n, burn.data(), n);  // the reference's third call (`cold`), then zeroed
    std::vector<float> a((size_t)m * lda, 0.f), b((size_t)k * ldb, 0.f), cold((size_t)m * ldc, 0.f),
        cref((size_t)m * ldc, 0.f);
    copy_matrix(m, k, dense_a.data(), k, a.data(), lda);
    copy_matrix(k, n, dense_b.data(), n, b.data(), ldb);

    // the oracle, timed with dclock() so the CPU baseline sits beside the GPU number
    double ref_gflops = 0.0;
    int ref_cores = 0;
    if (o.ref != "skip") {
      const double t0 = dclock();
      if (o.ref == "serial") {
        REF_MMult_serial(m, n, k, a.data(), lda, b.data(), ldb, cref.data(), ldc);
        ref_cores = 1;
      } else if (o.ref == "blas") {
        if (!REF_MMult_blas(m, n, k, a.data(), lda, b.data(), ldb, cref.data(), ldc)) {
          std::fprintf(stderr, "REF=blas: no host BLAS with cblas_sgemm could be loaded (set MMULT_BLAS_LIB)\n");
          return 2;
        }
        ref_cores = REF_MMult_threads();   // the BLAS's own threading; an upper bound
      } else {
        REF_MMult(m, n, k, a.data(), lda, b.data(), ldb, cref.data(), ldc);
        ref_cores = REF_MMult_threads();
      }
      ref_gflops = flops * 1e-9 / (dclock() - t0);
    }

    double seconds = 0.0;
    float phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (sharded) {
      // one call = h2d, ONE broadcast, NREPEATS back-to-back GEMM launches per device, d2h; the GEMM
      // phase is reported per launch (the reference times NREPEATS launches, cuda/test_MMult.cpp:98-118)
      // the three host arrays of this size are page-locked for the calls below (the role pinned staging plays
      // for any H2D copy that should run at the link's rate) and released before they are freed
      const bool pinned = mmh_shard_pin(shard, a.data(), a.size() * sizeof(float)) == MMH_OK &&
                          mmh_shard_pin(shard, b.data(), b.size() * sizeof(float)) == MMH_OK &&
                          mmh_shard_pin(shard, cold.data(), cold.size() * sizeof(float)) == MMH_OK;
      for (int rep = 0; rep < o.warmup; ++rep)
        MMH_CHECK(mmh_shard_sgemm(shard, m, n, k, a.data(), lda, b.data(), ldb, cold.data(), ldc, 1, nullptr));
      MMH_CHECK(mmh_shard_sgemm_streamed(shard, m, n, k, a.data(), lda, b.data(), ldb, cold.data(), ldc, o.nrepeats, o.b_chunks,
                                         phase_ms));
      seconds = phase_ms[2] * 1e-3;
      (void)pinned;
      (void)mmh_shard_unpin(shard, a.data());
      (void)mmh_shard_unpin(shard, b.data());
      (void)mmh_shard_unpin(shard, cold.data());
    } else if (cpu_only) {
      double best = 0.0;
      for (int rep = 0; rep < o.nrepeats; ++rep) {
        std::fill(cold.begin(), cold.end(), 0.f);
        const double t0 = dclock();
        REF_MMult_serial(m, n, k, a.data(), lda, b.data(), ldb, cold.data(), ldc);
        const double dt = dclock() - t0;
        best = rep == 0 ? dt : std::min(best, dt);
      }
      seconds = best;
    } else if (host_flavour) {
      // armv7/aarch64 convention: re-zero C, time each call with dclock, keep the best
      double best = 0.0;
      for (int rep = 0; rep < o.nrepeats + o.warmup; ++rep) {
        std::fill(cold.begin(), cold.end(), 0.f);
        const double t0 = dclock();
        MY_MMult(m, n, k, a.data(), lda, b.data(), ldb, cold.data(), ldc);
        const double dt = dclock() - t0;
        best = rep == 0 ? dt : std::min(best, dt);
      }
      seconds = best;
    } else {
      float *d_A, *d_B, *d_C;
      HIP_CHECK(hipMalloc(&d_A, a.size() * sizeof(float)));
      HIP_CHECK(hipMalloc(&d_B, b.size() * sizeof(float)));
      HIP_CHECK(hipMalloc(&d_C, cold.size() * sizeof(float)));
      HIP_CHECK(hipMemcpy(d_A, a.data(), a.size() * sizeof(float), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_B, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
      auto call = [&] {
        if (kid == kRocblas)
          MMH_CHECK(mmh_sgemm_rocblas(handle, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, nullptr));
        else if (kid == kHipblaslt)
          MMH_CHECK(mmh_sgemm_hipblaslt(handle, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, nullptr));
        else
          MY_MMult(handle, m, n, k, d_A, lda, d_B, ldb, d_C, ldc);
      };
      for (int rep = 0; rep < o.warmup; ++rep) call();
      if (o.warmup_ms > 0) {
        const double t_end = dclock() + o.warmup_ms * 1e-3;
        do {
          for (int rep = 0; rep < 8; ++rep) call();
          HIP_CHECK(hipDeviceSynchronize());
        } while (dclock() < t_end);
        // ... and then until the rate has SETTLED: bursts of NREPEATS launches, timed like the trials below, until two
        // in a row agree within 1 % (at most 12).  After seconds of host work (REF) the chip's clock sometimes needs
        // more than the fixed 50 ms: a point of the sustained sweep then read 10 % low in all three trials, in one
        // process out of a few, at a different size each time -- for the vendor libraries too (profiles/r03_notes.md
        // section 7); eight sweeps with REF=skip, where the GPU never idles, agree within 1.5 % at every size.
        float prev = 0.f;
        for (int burst = 0, agreed = 0; burst < 12 && agreed < 2; ++burst) {
          HIP_CHECK(hipEventRecord(start, nullptr));
          for (int rep = 0; rep < o.nrepeats; ++rep) call();
          HIP_CHECK(hipEventRecord(stop, nullptr));
          HIP_CHECK(hipEventSynchronize(stop));
          float ms = 0.f;
          HIP_CHECK(hipEventElapsedTime(&ms, start, stop));
          agreed = (prev > 0.f && std::fabs(ms - prev) <= 0.01f * prev) ? agreed + 1 : 0;
          prev = ms;
        }
      }
      std::vector<float> trial_ms;
      for (int trial = 0; trial < o.trials; ++trial) {
        HIP_CHECK(hipEventRecord(start, nullptr));
        for (int rep = 0; rep < o.nrepeats; ++rep) call();
        HIP_CHECK(hipEventRecord(stop, nullptr));
        HIP_CHECK(hipEventSynchronize(stop));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, start, stop));
        trial_ms.push_back(ms);
      }
      std::sort(trial_ms.begin(), trial_ms.end());
      seconds = trial_ms[trial_ms.size() / 2] * 1e-3 / o.nrepeats;
      HIP_CHECK(hipMemcpy(cold.data(), d_C, cold.size() * sizeof(float), hipMemcpyDeviceToHost));
      HIP_CHECK(hipFree(d_A));
      HIP_CHECK(hipFree(d_B));
      HIP_CHECK(hipFree(d_C));
    }
    const double gflops = flops * 1e-9 / seconds;

    double diff = -1.0;
    if (o.ref != "skip") {
      diff = compare_matrices(m, n, cold.data(), ldc, cref.data(), ldc);
      if (diff > 0.5 || diff < -0.5) {
        std::printf("diff too big !\n");
        return -1;
      }
    }
    if (o.extended && sharded)
      std::printf("%d %.2f %le %.2f %.3f %.3f %.3f %.3f %.3f %d \n", p, gflops, diff,
                  100.0 * gflops / (o.ngpus * kPeakTflops * 1e3), phase_ms[0], phase_ms[1], phase_ms[2],
                  phase_ms[3], phase_ms[4], (int)phase_ms[5]);
    else if (o.extended)
      std::printf("%d %.2f %le %.2f %.3f %d \n", p, gflops, diff,
                  100.0 * gflops / (kPeakTflops * 1e3), ref_gflops, ref_cores);
    else
      std::printf("%d %.2f %le \n", p, gflops, diff);
    std::fflush(stdout);
    if (json_out) {
      std::string launched = cpu_only ? "serial triple loop (no GPU)" : mmh_last_launch();
      for (auto &ch : launched)
        if (ch == '"' || ch == '\\') ch = '\'';
      std::fprintf(json_out,
                   "%s\n {\"p\": %d, \"m\": %d, \"n\": %d, \"k\": %d, \"gflops\": %.2f, \"diff\": %.6e, "
                   "\"pct_of_fp32_mfma_peak\": %.2f, \"seconds\": %.6e, \"flavour\": \"%s\", \"kernel\": \"%s\", "
                   "\"launched\": \"%s\"}",
                   json_rows++ ? "," : "", p, m, n, k, gflops, diff, 100.0 * gflops / (kPeakTflops * 1e3), seconds,
                   o.flavour.c_str(), o.kernel.c_str(), launched.c_str());
      std::fflush(json_out);
    }
  }
  if (json_out) {
    std::fprintf(json_out, "\n]\n");
    std::fclose(json_out);
  }

  if (shard) MMH_CHECK(mmh_shard_destroy(shard));
  if (handle) MMH_CHECK(mmh_destroy(handle));
  std::printf("];\n");
  return 0;
}
