# dev: ablation builds of the halo kernel (stores / MFMAs / loads switched off one at a time; built beforehand with
# scripts/build_dev_lib.sh <v> "-DHALO_DBG_<v> -DRFX_HALO_ONLY_R3" gemm_fwd_halo.hip), timed on the decoder rewrite layers
python scripts/perf_halo.py 5 2>/dev/null | head -4
for v in NO_EPI NO_MFMA NO_BLOAD NO_ALOAD; do
  echo "== $v"; PERF_FWD_ONLY=1 RFX_LIBPATH_DEV=$PWD/remfx_amd/_C/libremfx_hip_$v.so python scripts/perf_halo.py 5 2>&1 | grep -v amdgpu.ids | head -12
done
