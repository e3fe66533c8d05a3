#!/bin/bash
# round 3, call 25: last validation at HEAD -- the GPU suite twice, smoke, the default bench line, one more fuzz seed
set -u
O=gpurun_out/r03u; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  timeout 600 python -m pytest tests -m gpu -q > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -E 'passed|failed|Aborted' $O/pytest_$i.log | tail -2 | tr '\n' ' ')"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel'][:40], d['cpu_baseline']['value'], d['extras']['int8_4096_tops'])"
timeout 400 python tools/fuzz.py 250 20 21 2>&1 | tail -2
