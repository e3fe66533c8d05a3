#!/bin/bash
# round 3, call 18: how often, and where, does a sustained sweep point read low?  Eight full sweeps (REF=skip), side by side.
set -u
O=gpurun_out/r03p; mkdir -p $O
H=how-to-optimize-gemm_amd/harness
for i in 1 2 3 4 5 6 7 8; do
  ( cd $H && KERNEL=auto REF=skip WARMUP_MS=50 TRIALS=3 timeout 200 ./test_MMult.x ) 2>/dev/null | awk 'NF>=3 && $1+0>0{print $2}' > $O/s$i.txt
done
( cd $H && KERNEL=auto REF=skip WARMUP_MS=50 TRIALS=3 timeout 200 ./test_MMult.x ) 2>/dev/null | awk 'NF>=3 && $1+0>0{print $1}' > $O/n.txt
paste $O/n.txt $O/s1.txt $O/s2.txt $O/s3.txt $O/s4.txt $O/s5.txt $O/s6.txt $O/s7.txt $O/s8.txt | awk '{mx=0; mn=1e9; for(i=2;i<=NF;i++){if($i>mx)mx=$i; if($i<mn)mn=$i}; printf "%d", $1; for(i=2;i<=NF;i++) printf " %.1f", $i/1000; printf "  | min/max %.3f\n", mn/mx}' | tee $O/eight_sweeps.txt
