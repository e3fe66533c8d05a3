#!/bin/bash
# round 3, call 17: the round's early measurements again (their files were lost with the container): the LDS-DMA
# alignment / range probe, round-2 library vs this one on the stream-K sizes, the int8 leading-dimension probe
set -u
O=gpurun_out/r03o; mkdir -p $O
tools/probes/lds_dma_align_probe.x > $O/probe_align.txt 2>&1; tail -8 $O/probe_align.txt
timeout 300 python tools/ab_r02.py > $O/ab_r02.txt 2>&1; cut -c1-160 $O/ab_r02.txt | tail -16
timeout 300 python tools/i8_ld_probe.py 0 > $O/i8_ld_probe.txt 2>&1; tail -12 $O/i8_ld_probe.txt
