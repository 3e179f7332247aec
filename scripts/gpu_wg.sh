#!/bin/bash
mkdir -p gpurun_out/wg
timeout 300 python scripts/perf_wgrad.py 2>&1 | grep -v amdgpu.ids > gpurun_out/wg/le.txt; cat gpurun_out/wg/le.txt
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -3
