#!/bin/bash
mkdir -p gpurun_out/fft
timeout 900 python -m pytest tests/test_gpu_stft.py -x -q -m gpu > gpurun_out/fft/t.log 2>&1
tail -15 gpurun_out/fft/t.log
for nb in 1 2 4; do echo "== nb $nb"; RFX_FFT_NB=$nb timeout 300 python scripts/perf_fft.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/fft/perf.txt 2>&1
cat gpurun_out/fft/perf.txt
