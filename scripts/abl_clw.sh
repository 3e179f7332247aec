#!/bin/bash
# dev: ablation builds of the channels-last weight-gradient kernel (python scripts/build_abl.py cl_wgrad RFX_CLW_DBG_BUILD <n> ...;
# bits: 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no barrier, 16 no staggering)
export PERF_CL_LAYERS=${1:-48}
for v in ${2:-0 16 1 2 4 3 6}; do
  echo "== RFX_CLW_DBG_BUILD=$v"
  L=""; [ $v != 0 ] && L=$PWD/remfx_amd/_C/abl/lib_cl_wgrad_$v.so
  RFX_LIBPATH_DEV=$L python scripts/perf_clw.py 2>&1 | grep "3x3\|k8s4\|1x1\|conv_tr"
done
