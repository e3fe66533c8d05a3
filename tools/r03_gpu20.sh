#!/bin/bash
# round 3, call 20: the GPU suite three times in a row (one run in four aborted before the pinned-host test got its own process)
set -u
O=gpurun_out/r03q; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python -m pytest tests -m gpu -q > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -E 'passed|failed|Aborted' $O/pytest_$i.log | tail -2 | tr '\n' ' ')"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
