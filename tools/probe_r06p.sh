set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
timeout 1200 python tools/tile_sweep.py --ab --sizes 1024:4096:128 --variants auto,mfma_128x128_dma5/sk2,exp5_128x128_rs0/sk2,mfma_128x64_dma5/sk2,exp5_128x64_rs0/sk2,mfma_64x64_dma5/sk2,exp5_64x64_rs0/sk2,mfma_96x96_dma5,exp5_96x96_rs0,exp5_160x160_l4,exp5_160x96_l1d2 --out $O/ts_rs --check --rounds 3 > $O/ts_rs.txt 2>&1
tail -5 $O/ts_rs.txt
cat $O/ts_rs.md
