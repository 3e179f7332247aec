#!/bin/bash
# same-box A/B of environment switches: each `run` line is one configuration;  B = per-GPU batch (default 64)
B=${B:-64}
run() { env "$@" python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('B=$B $*', j['ms_per_step'])"; }
run RFX_LSTM_LOCAL=1
run RFX_LSTM_LOCAL=0
run RFX_LSTM_LOCAL=1
run RFX_LSTM_LOCAL=0
run RFX_LSTM_LOCAL=1 RFX_LSTM_LOCAL_BWD=0
