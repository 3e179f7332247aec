#!/bin/bash
mkdir -p gpurun_out/r3r
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_hdemucs.py tests/test_gpu_bf16_mixed.py tests/test_gpu_dconv_fused.py tests/test_gpu_bf16x3.py -x -q -m gpu > gpurun_out/r3r/t.log 2>&1
tail -4 gpurun_out/r3r/t.log

