#!/bin/bash
# measurement only: upper bound of what the channel-major DConv branches of width C cost (branch skipped = identity)
mkdir -p gpurun_out/skip
for c in "" 192 384 192,384; do
  RFX_DBG_SKIP_DCONV_C=$c python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/skip/b_$c.json 2> gpurun_out/skip/b_$c.err
  python - <<PY
import json
j=json.load(open("gpurun_out/skip/b_$c.json"))
print("skip=[$c]", j["ms_per_step"])
PY
done
