#!/bin/bash
mkdir -p gpurun_out/s1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/probes/unaligned.hip -o /tmp/unal 2> gpurun_out/s1/unal.err && /tmp/unal > gpurun_out/s1/unal.txt 2>&1
timeout 600 python scripts/fill_sites.py 64 > gpurun_out/s1/fill.txt 2> gpurun_out/s1/fill.err
tail -70 gpurun_out/s1/fill.txt; cat gpurun_out/s1/unal.txt
