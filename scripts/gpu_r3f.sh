#!/bin/bash
mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_effects.py -x -q > gpurun_out/r3f/fx.log 2>&1; tail -15 gpurun_out/r3f/fx.log
python -m pytest tests/test_gpu_train_script.py -x -q > gpurun_out/r3f/scripts.log 2>&1; tail -12 gpurun_out/r3f/scripts.log
for m in side off side; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --sink $m 2>> gpurun_out/r3f/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['config']['final_loss'])" | tee -a gpurun_out/r3f/ab.txt
done
python - <<'P' > gpurun_out/r3f/fxtime.txt 2>&1
import torch, time
from remfx_amd import effects as E
x = torch.randn(64, 1, 262144, device="cuda") * 0.1
mods = [E.RandomPedalboardDistortion(48000), E.RandomPedalboardDelay(48000), E.RandomPedalboardChorus(48000), E.RandomPedalboardCompressor(48000), E.RandomPedalboardReverb(48000), E.LoudnessNormalize(48000, -20)]
for m in mods:
    m(x); torch.cuda.synchronize(); t = time.time()
    for _ in range(3): m(x)
    torch.cuda.synchronize(); print(type(m).__name__, "64 x 262144: %.2f ms" % ((time.time() - t) / 3 * 1e3))
P
cat gpurun_out/r3f/fxtime.txt
