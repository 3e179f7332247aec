#!/bin/bash
# cl_conv ablation builds (scripts/build_abl.py cl_conv_m_glu RFX_CLC_DBG_BUILD ...: 1 no DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue, 16 no barrier)
for d in ${ABL:-0 8 1 9 4 12}; do
  echo "== RFX_CLC_DBG_BUILD=$d"
  L=""
  [ $d != 0 ] && L=$PWD/remfx_amd/_C/abl/lib_cl_conv_m_glu_$d.so
  RFX_LIBPATH_DEV=$L PERF_CL_LAYERS=${PERF_CL_LAYERS:-48} python scripts/perf_cl.py 64 2>&1 | grep "glu"
done
