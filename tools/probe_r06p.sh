set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
timeout 1200 python tools/tile_sweep.py --ab --sizes 1024:2048:128 --variants auto,mfma_64x64_dma5/sk1,exp5_64x64_d3/sk1,exp5_64x64_d1/sk1,mfma_96x64_dma5,exp5_96x64_d3,mfma_128x64_dma5/sk1,exp5_128x64_d3/sk1 --out $O/ts_d3 --check --rounds 5 > $O/ts_d3.txt 2>&1
cat $O/ts_d3.md
timeout 600 python tools/tile_sweep.py --ab --shapes "4096,4096,4096;3072,3072,3072" --variants auto,mfma_128x64_dma5/sk1,exp5_128x64_d3/sk1,mfma_96x64_dma5,exp5_96x64_d3 --out $O/ts_d3b --check --rounds 5 > $O/ts_d3b.txt 2>&1
cat $O/ts_d3b.md
