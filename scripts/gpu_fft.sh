#!/bin/bash
mkdir -p gpurun_out/fft
timeout 900 python -m pytest tests/test_gpu_stft.py tests/test_gpu_fullsize_properties.py tests/test_gpu_umx.py -x -q -m gpu > gpurun_out/fft/t.log 2>&1
tail -4 gpurun_out/fft/t.log
timeout 300 python scripts/perf_fft.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fft/perf.txt; cat gpurun_out/fft/perf.txt
