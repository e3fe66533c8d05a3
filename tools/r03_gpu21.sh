#!/bin/bash
# round 3, call 21: the new GPU test (host plan vs launch), differential fuzz of every kernel variant with the final library (three seeds), int8 fuzz
set -u
O=gpurun_out/r03r; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -k "host_plan or rim or shard" 2>&1 | tail -3
for seed in 11 12 13; do timeout 400 python tools/fuzz.py 250 20 $seed > $O/fuzz_$seed.txt 2>&1; echo "fuzz seed $seed: $(tail -1 $O/fuzz_$seed.txt)"; done
timeout 300 python tools/fuzz_i8.py > $O/fuzz_i8.txt 2>&1; echo "fuzz_i8: $(tail -1 $O/fuzz_i8.txt)"
