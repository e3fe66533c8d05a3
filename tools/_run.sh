cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --force-shard --steps 5 --warmup 2 2>gpurun_out/shard_stderr.log | tail -1 | cut -c1-1400
timeout 600 python bench.py 2>gpurun_out/bench_stderr.log | tail -1 > gpurun_out/bench_last.json; python -c "
import json; d=json.load(open('gpurun_out/bench_last.json')); print(d['value'], d['roofline']['frac'], d['extras'].get('int8_4096_tops'), d['extras'].get('probe_mfma_i8_tops_random_operands'), d['cpu_baseline'])"
