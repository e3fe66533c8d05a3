cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "int8 or quantised or fuzz" 2>&1 | tail -4
I8_MODES=3,4,0 timeout 300 python tools/misc_bench.py i8 2>&1 | tail -6
