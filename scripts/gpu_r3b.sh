#!/bin/bash
mkdir -p gpurun_out/r3b
export RFX_TOL_LOG=gpurun_out/r3b/tol.jsonl
( python -m pytest tests/test_gpu_conv.py -k tcn_full -q -s
  python -m pytest tests/test_gpu_classifier_chain.py tests/test_gpu_dcunet.py -q -s ) > gpurun_out/r3b/tests.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r3b/full.log 2>&1
grep -n "passed\|failed" gpurun_out/r3b/tests.log gpurun_out/r3b/full.log
