#!/bin/bash
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_dptnet.py -x -q -s > gpurun_out/r3j/t.log 2>&1; tail -25 gpurun_out/r3j/t.log
RFX_TEST_MODES=bf16 timeout 600 python -m pytest tests/test_gpu_classifier_chain.py -x -q > gpurun_out/r3j/t2.log 2>&1; tail -3 gpurun_out/r3j/t2.log
