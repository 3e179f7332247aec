#!/bin/bash
# dev: second build of the library with extra -D flags on selected translation units, for same-box A/B through RFX_LIBPATH_DEV
#   bash scripts/build_dev_lib.sh <tag> "<flags>" file1.hip [file2.hip ...]   -> remfx_amd/_C/libremfx_hip_<tag>.so
set -e
TAG=$1; FLAGS=$2; shift 2
C=remfx_amd/_C; mkdir -p $C/dev_$TAG
OBJS=""
for o in $C/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" == "$b" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Iinclude -Iremfx_amd/csrc $FLAGS -c remfx_amd/csrc/$b.hip -o $C/dev_$TAG/$b.o
      use=$C/dev_$TAG/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libremfx_hip_$TAG.so $OBJS
echo built $C/libremfx_hip_$TAG.so
