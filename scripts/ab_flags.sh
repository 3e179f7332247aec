#!/bin/bash
# same-box A/B of environment switches: each `run` line is one configuration
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$*', j['ms_per_step'])"; }
run A=0
run RFX_CL_BM96_K=0 RFX_CL_BM96_NTC=0
run RFX_CL_BM96_K=0 RFX_CL_BM96_NTC=1
run RFX_CL_BM96_K=800 RFX_CL_BM96_NTC=0
run RFX_CL_BM96_K=400 RFX_CL_BM96_NTC=0
run RFX_CL_BM96_K=0 RFX_CL_BM96_NTC=0
run A=0
