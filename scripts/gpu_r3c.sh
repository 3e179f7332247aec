#!/bin/bash
mkdir -p gpurun_out/r3c
python bench.py --steps 10 --warmup 3 > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
python __graft_entry__.py smoke > gpurun_out/r3c/smoke.log 2>&1
python -m pytest tests/test_gpu_hdemucs.py tests/test_gpu_conv.py -x -q > gpurun_out/r3c/t.log 2>&1
tail -3 gpurun_out/r3c/smoke.log gpurun_out/r3c/t.log; tail -3 gpurun_out/r3c/bench.err
