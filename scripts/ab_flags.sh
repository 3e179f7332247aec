#!/bin/bash
# same-box A/B of environment switches: each `run` line is one configuration
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', j['ms_per_step'])"; }
run A=0
run RFX_CLD_GRID_FWD=512
run RFX_CLD_GRID_FWD=768
run RFX_CLW_PW=64
run RFX_CLW_AHEAD=3
run A=0
