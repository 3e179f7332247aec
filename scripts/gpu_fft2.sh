#!/bin/bash
mkdir -p gpurun_out/fft
: > gpurun_out/fft/dbg.txt
for d in 0 1 2 4 8 7 15; do for nb in 1 4; do echo "== dbg $d nb $nb" >> gpurun_out/fft/dbg.txt; RFX_FFT_NB=$nb RFX_FFT_DBG=$d timeout 300 python scripts/perf_fft.py 2>&1 | grep analysis >> gpurun_out/fft/dbg.txt; done; done
cat gpurun_out/fft/dbg.txt
