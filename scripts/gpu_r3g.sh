#!/bin/bash
mkdir -p gpurun_out/r3g
timeout 600 python -m pytest tests/test_gpu_dconv_fused.py -x -q -s > gpurun_out/r3g/t.log 2>&1; tail -40 gpurun_out/r3g/t.log
