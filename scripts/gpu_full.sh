#!/bin/bash
mkdir -p gpurun_out/full
python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1; tail -6 gpurun_out/full/pytest.log
python __graft_entry__.py smoke 2>&1 | tail -3
