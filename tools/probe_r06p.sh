set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06t; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
TAG=r06t STEPS="dataset" DATASETS="fit heldout" bash tools/gpu_call.sh > $O/dataset_pass.log 2>&1
tail -4 $O/dataset_pass.log | cut -c1-200
