set -u
OUT=gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout 900 python tools/smalln_sweep.py --rounds 3 --sizes $(seq -s, 1024 128 4096) --variants auto,rocblas,hipblaslt > $OUT/sweep_vs_vendor.md 2> $OUT/sweep_vs_vendor.err
grep -v "^<" $OUT/sweep_vs_vendor.md | tail -26
