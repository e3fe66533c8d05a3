"""The C++ host side (how-to-optimize-gemm_amd/harness): builds, exports the
reference's MY_MMult symbol, reproduces the driver's stdout contract.  CPU
tests use FLAVOUR=cpu (BASELINE.json config 1: MY_MMult := the triple loop,
N=256, no GPU); GPU tests run the real sweep and the reference's own driver
linked against our MY_MMult."""
import os
import re
import subprocess

import pytest

from conftest import REPO

HARNESS = os.path.join(REPO, "how-to-optimize-gemm_amd", "harness")
EXE = os.path.join(HARNESS, "test_MMult.x")
DROPIN_AARCH64 = os.path.join(REPO, "oracle", "_ref", "test_MMult_dropin_aarch64.x")
DROPIN = os.path.join(REPO, "oracle", "_ref", "test_MMult_dropin.x")
DROPIN_VULKAN = os.path.join(REPO, "oracle", "_ref", "test_MMult_dropin_vulkan.x")
ROW = re.compile(r"^(\d+) (\d+\.\d+) (-?\d\.\d+e[+-]\d+) $")          # cuda flavour: %d %.2f %le
ROW_LE = re.compile(r"^(\d+) (\d\.\d+e[+-]\d+) (-?\d\.\d+e[+-]\d+) $")  # armv7 flavour: %d %le %le


def build():
    subprocess.check_call(["make", "-s", "-C", HARNESS, "all"])


def run(env_extra, exe=EXE, timeout=900):
    env = dict(os.environ)
    env.update({k: str(v) for k, v in env_extra.items()})
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=timeout)
    return r.returncode, r.stdout, r.stderr


def parse(stdout, row=ROW):
    """cuda/plot.py:5-28's reading of an output file body: header lines, then
    'p gflops diff ' rows until a line with <= 2 tokens."""
    lines = stdout.splitlines()
    i = lines.index("MY_MMult = [")
    rows = []
    for ln in lines[i + 1:]:
        if ln == "];":
            break
        mobj = row.match(ln)
        assert mobj, f"row does not match '%d %.2f %le ': {ln!r}"
        rows.append((int(mobj.group(1)), float(mobj.group(2)), float(mobj.group(3))))
    assert lines[-1] == "];"
    return rows


def test_builds_and_exports_reference_symbol():
    build()
    out = subprocess.check_output(["nm", os.path.join(HARNESS, "MMult_hip.o")], text=True)
    assert " T _Z8MY_MMultiiiPfiS_iS_i" in out            # 9-arg host flavour, C++ linkage
    assert " T _Z8MY_MMultiiiPfS_S_" in out                # 6-arg vulkan-directory flavour (returns ms)


def test_cpu_plumbing_config1():
    """configs[0]: REF_MMult triple loop, N=256 square, CPU only -> diff = 0."""
    build()
    rc, out, err = run({"FLAVOUR": "cpu", "PFIRST": 256, "PLAST": 256, "NREPEATS": 2, "REF": "serial"})
    assert rc == 0, err
    rows = parse(out)
    assert [r[0] for r in rows] == [256] and rows[0][2] == 0.0 and rows[0][1] > 0.05


def test_json_sidecar_beside_the_reference_format(tmp_path):
    """JSON=<path>: stdout keeps the reference's format; the same rows land in a JSON array with the
    shape, the rate as % of the fp32 MFMA peak and what was launched."""
    import json
    build()
    path = tmp_path / "sweep.json"
    rc, out, err = run({"FLAVOUR": "cpu", "PFIRST": 64, "PLAST": 192, "PINC": 64, "NREPEATS": 1, "JSON": str(path)})
    assert rc == 0, err
    rows = parse(out)
    side = json.load(open(path))
    assert [r["p"] for r in side] == [r[0] for r in rows] == [64, 128, 192]
    for r, s_ in zip(rows, side):
        assert abs(s_["gflops"] - r[1]) < 0.011 and s_["diff"] == r[2] == 0.0
        assert s_["m"] == s_["n"] == s_["k"] == s_["p"] and s_["flavour"] == "cpu" and "serial" in s_["launched"]


def test_cpu_sweep_format_and_threaded_ref_identical():
    build()
    rc, out, _ = run({"FLAVOUR": "cpu", "PFIRST": 40, "PLAST": 200, "PINC": 40, "NREPEATS": 1,
                      "INPUT": "seed:7"})
    assert rc == 0
    rows = parse(out)
    assert [r[0] for r in rows] == [40, 80, 120, 160, 200]
    # cref came from the row-parallel REF, cold from the serial loop: bit-identical
    assert all(r[2] == 0.0 for r in rows)
    # non-square and padded leading dimensions through the option parser
    rc, out, _ = run({"FLAVOUR": "cpu", "PFIRST": 64, "PLAST": 64, "M": 96, "K": 80, "LDA": 88,
                      "LDB": 72, "LDC": 100, "NREPEATS": 1, "INPUT": "mod3", "EXTENDED": 1})
    assert rc == 0 and "64 " in out


def test_blas_order_oracle_reproduces_the_published_diff_magnitudes():
    """the cuda directory's REF_MMult is cblas_sgemm (cuda/REF_MMult.cpp:9-13), and the
    diff column of every published cuda/output_MMult_cuda_*.m is a k-ordered fp32 chain against THAT
    blocked summation order: 7.2e-5 at p=1024, 1.07e-4 at 1280 (cuda/output_MMult_cuda_12.m:5,7).
    Here MY_MMult := the serial triple loop (FLAVOUR=cpu, the same chain up to FMA contraction) and
    REF=blas := the host BLAS found at run time; the diff must land in the published band (an
    order-of-magnitude pin: it depends on the BLAS build and the CPU)."""
    build()
    rc, out, err = run({"FLAVOUR": "cpu", "PFIRST": 1024, "PLAST": 1280, "PINC": 256, "NREPEATS": 1, "REF": "blas"},
                       timeout=600)
    assert rc == 0, err
    rows = parse(out)
    assert [r[0] for r in rows] == [1024, 1280]
    published = {1024: 7.247925e-05, 1280: 1.068115e-04}
    for p, _, diff in rows:
        assert published[p] / 4 <= diff <= published[p] * 4, (p, diff)
    # and the two oracles agree with each other far inside the harness tolerance (0.5)
    assert all(d < 1e-3 for _, _, d in rows)


@pytest.mark.gpu
def test_gpu_sweep_against_the_blas_order_oracle():
    """The same column with the GPU behind MY_MMult: the reference's own published magnitudes
    (cuda/output_MMult_cuda_12.m:5,13: 7.2e-5 at 1024, 1.6e-4 at 2048) within a factor of 4."""
    build()
    rc, out, err = run({"PFIRST": 1024, "PLAST": 2048, "PINC": 1024, "NREPEATS": 3, "REF": "blas"})
    assert rc == 0, err + out
    rows = parse(out)
    published = {1024: 7.247925e-05, 2048: 1.564026e-04}
    for p, _, diff in rows:
        assert published[p] / 4 <= diff <= published[p] * 4, (p, diff)


@pytest.mark.gpu
def test_device_flavour_sweep_on_gpu():
    build()
    rc, out, err = run({"PFIRST": 1024, "PLAST": 1536, "PINC": 256, "INPUT": "seed:11", "WARMUP": 2})
    assert rc == 0, err + out
    assert out.startswith('GPU Device 0: "')
    rows = parse(out)
    assert [r[0] for r in rows] == [1024, 1280, 1536]
    for p, gflops, diff in rows:
        assert 0.0 <= diff <= 2e-7 * p + 1e-6        # vs the unfused triple loop
        assert gflops > 5000
    # known-answer inputs: exactly zero, every kernel on the ladder
    for kern in ("mfma", "mfma256", "mfma_pipe", "mfma_simple", "valu", "valu_64x64", "naive", "rocblas", "mfma_splitk"):
        rc, out, err = run({"PFIRST": 1024, "PLAST": 1024, "INPUT": "mod3", "KERNEL": kern, "NREPEATS": 3})
        assert rc == 0, kern + err
        assert parse(out)[0][2] == 0.0, kern


@pytest.mark.gpu
def test_host_flavour_on_gpu():
    build()
    rc, out, err = run({"FLAVOUR": "host", "PFIRST": 48, "PLAST": 480, "PINC": 144, "INPUT": "seed:3",
                        "NREPEATS": 3})
    assert rc == 0, err
    rows = parse(out)
    assert [r[0] for r in rows] == [48, 192, 336, 480]
    assert all(0.0 <= r[2] <= 1e-4 for r in rows)


@pytest.mark.gpu
def test_sharded_flavour_single_process_on_gpu():
    """BASELINE config 4 through the C++ harness: mmh_sgemm_sharded with NGPUS=1 here (the
    RCCL broadcast is skipped for one device); exact on the known-answer inputs."""
    build()
    rc, out, err = run({"FLAVOUR": "sharded", "NGPUS": 1, "PFIRST": 512, "PLAST": 1024, "PINC": 512,
                        "INPUT": "mod3", "NREPEATS": 2})
    assert rc == 0, err
    rows = parse(out)
    assert [r[0] for r in rows] == [512, 1024] and all(r[2] == 0.0 for r in rows)
    # more devices than the box has: the harness exits non-zero (MMH_CHECK), it does not shrink the job
    import torch
    rc, out, err = run({"FLAVOUR": "sharded", "NGPUS": torch.cuda.device_count() + 1, "PFIRST": 512, "PLAST": 512})
    assert rc != 0 and "fewer visible devices" in err


@pytest.mark.gpu
def test_reference_driver_linked_against_our_MY_MMult():
    """The reference's own armv7/test_MMult.c + REF_MMult.c + compare_matrices.c
    (object code built from /root/reference by oracle/Makefile) with ONLY
    MY_MMult replaced by ours: its (j-i)%2 inputs are integer-valued, so its
    diff column must be exactly 0 over its whole sweep 40..700 step 40."""
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/test_MMult_dropin.x was not built (no /root/reference at build time)")
    rc, out, err = run({}, exe=DROPIN)
    assert rc == 0, err
    rows = parse(out, ROW_LE)
    assert [r[0] for r in rows] == list(range(40, 701, 40))
    assert all(r[2] == 0.0 for r in rows)


@pytest.mark.gpu
def test_reference_aarch64_driver_linked_against_our_MY_MMult():
    """The reference's aarch64/test_MMult.cpp + REF_MMult.cpp + compare_matrices.cpp (row-major,
    lda = k, all-ones inputs, C zeroed before each of its 10 calls) with ONLY MY_MMult replaced by
    ours: every C element is exactly k, so its diff column is 0 over its sweep 48..960 step 48."""
    if not os.path.exists(DROPIN_AARCH64):
        pytest.skip("oracle/_ref/test_MMult_dropin_aarch64.x was not built (no /root/reference at build time)")
    rc, out, err = run({}, exe=DROPIN_AARCH64)
    assert rc == 0, err
    rows = parse(out, ROW_LE)
    assert [r[0] for r in rows] == list(range(48, 961, 48))
    assert all(r[2] == 0.0 for r in rows)


@pytest.mark.gpu
def test_reference_vulkan_driver_linked_against_our_MY_MMult():
    """The reference's vulkan/test_MMult.cpp + REF_MMult.cpp + compare_matrices.cpp + random_matrix.cpp
    (plain C++: `float MY_MMult(m, n, k, a, b, c)` returns milliseconds, drand48 inputs in [-1, 1),
    sweep 64..512 step 64) with ONLY MY_MMult replaced by ours.  oracle/Makefile compiles the
    reference's files with -mfma -ffp-contract=fast -- the REF loop as FMA hardware runs it -- so the
    diff column is EXACTLY 0 on random inputs, and the GFLOPS column comes from the milliseconds our
    MY_MMult returned (the GEMM's device time)."""
    if not os.path.exists(DROPIN_VULKAN):
        pytest.skip("oracle/_ref/test_MMult_dropin_vulkan.x was not built (no /root/reference at build time)")
    rc, out, err = run({}, exe=DROPIN_VULKAN)
    assert rc == 0, err
    rows = parse(out, ROW)        # vulkan/test_MMult.cpp:82-83 prints '%d %.2f %le '
    assert [r[0] for r in rows] == list(range(64, 513, 64))
    assert all(r[2] == 0.0 for r in rows)
    assert all(r[1] > 0.0 for r in rows)      # a real device time came back from every call


@pytest.mark.gpu
def test_probes_go_to_stderr_and_leave_the_result_format_alone():
    build()
    rc, out, err = run({"PFIRST": 256, "PLAST": 256, "PROBES": 1, "REF": "serial"})
    assert rc == 0, err
    rows = parse(out)
    assert [r[0] for r in rows] == [256]
    mobj = re.search(r"probes: mfma_f32 (\d+\.\d) TFLOP/s, hbm copy (\d+) GB/s, hbm read (\d+) GB/s, "
                     r"lds read (\d+) GB/s \((\d+\.\d) B/clk/CU", err)
    assert mobj, err
    assert 100 < float(mobj.group(1)) < 170 and 3000 < float(mobj.group(2)) < 8000
    assert 100 < float(mobj.group(5)) <= 260      # the roof of ds_read_b128 is 256 B/clk/CU
