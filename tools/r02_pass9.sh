mkdir -p gpurun_out/r02h
python -m pytest tests/test_harness.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r02h/pytest_harness.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vulkan or lds_probe or restore or host" 2>&1 | tail -5 > gpurun_out/r02h/pytest_new.log
python - > gpurun_out/r02h/lds_probe.txt 2>&1 <<'P'
import how_to_optimize_gemm_amd as H
with H.MMult(0, "auto") as mm:
    for r in range(3):
        print({w: round(mm.probe_lds_read(w), 0) for w in (16, 8, 4, -8)})
P
oracle/_ref/test_MMult_dropin_vulkan.x > gpurun_out/r02h/dropin_vulkan.txt 2>&1
PROBES=1 PFIRST=1024 PLAST=1024 REF=skip how-to-optimize-gemm_amd/harness/test_MMult.x > gpurun_out/r02h/probes.txt 2>&1
