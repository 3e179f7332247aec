#!/bin/bash
# dev: same-box A/B of two builds of the library (old = remfx_amd/_C/old_libremfx_hip.so)
mkdir -p gpurun_out/ab; : > gpurun_out/ab/ab.txt
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export RFX_LIBPATH_DEV=$PWD/remfx_amd/_C/old_libremfx_hip.so; else unset RFX_LIBPATH_DEV; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-also $EXTRA > gpurun_out/ab/b.json 2>> gpurun_out/ab/err.txt
    python - "$v" <<'P' >> gpurun_out/ab/ab.txt
import json,sys
d=json.loads(open("gpurun_out/ab/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["config"].get("final_loss"))
P
  done
done
cat gpurun_out/ab/ab.txt
