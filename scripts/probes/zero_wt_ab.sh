# Root-cause A/B of the fill + atomics hazard: the failing configuration of round 5 (GroupNorm statistics from the GEMM
# epilogue: fp64 atomics into a buffer rfx_zero has just filled, time branch on its own stream) with the plain fill and with the
# write-through fill.   bash scripts/probes/zero_wt_ab.sh [reps]
N=${1:-40}
mkdir -p gpurun_out/r06
for wt in 0 1; do
  RFX_DEV=1 RFX_DCONV_EPI_STATS=1 RFX_ZERO_WT=$wt python scripts/probes/batch_invariance_loop.py $N > gpurun_out/r06/zero_wt_$wt.txt 2>&1
  echo "RFX_ZERO_WT=$wt: bad reps (rms err / scale > 1e-5):" $(awk '/rms err/ {split($0,a,"= "); split(a[2],b," "); if (b[1]+0 > 1e-5) n++} END {print n+0}' gpurun_out/r06/zero_wt_$wt.txt) of $N
done
