#!/bin/bash
mkdir -p gpurun_out/fft
: > gpurun_out/fft/ab.txt
echo "== old" >> gpurun_out/fft/ab.txt
RFX_LIBPATH_DEV=$PWD/remfx_amd/_C/old_libremfx_hip.so timeout 300 python scripts/perf_fft.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/fft/ab.txt
for nb in 1 2 4; do echo "== new nb $nb" >> gpurun_out/fft/ab.txt; RFX_FFT_NB=$nb timeout 300 python scripts/perf_fft.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/fft/ab.txt; done
echo "== new nb1 dbg15" >> gpurun_out/fft/ab.txt; RFX_FFT_NB=1 RFX_FFT_DBG=15 timeout 300 python scripts/perf_fft.py 2>&1 | grep analysis >> gpurun_out/fft/ab.txt
cat gpurun_out/fft/ab.txt
