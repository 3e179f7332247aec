#!/bin/bash
# round-3 first GPU pass: new parity tests + baseline bench
mkdir -p gpurun_out/r3a
export RFX_TOL_LOG=gpurun_out/r3a/tol.jsonl
( python -m pytest tests/test_gpu_conv.py -k tcn_full -x -q -s
  python -m pytest tests/test_gpu_fullsize_properties.py -k "batching or b64" -x -q -s
  python -m pytest tests/test_gpu_hdemucs.py -k localstate -x -q
  python -m pytest tests/test_gpu_classifier_chain.py tests/test_gpu_dcunet.py -x -q -s ) > gpurun_out/r3a/tests.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -5 gpurun_out/r3a/tests.log; cat gpurun_out/r3a/bench.json | cut -c1-400
