# Per-launch time of the cross-block kernel with parts of its work compiled out (TSM_C31_X bit mask, wrong results by design):
# which stream bounds the launch.  Libraries libtsm_hip_x<mask>.so are built beforehand (TSM_BUILD_DEFS=-DTSM_C31_X=<mask>).
R=$(pwd)
export TSM_FUSE_C3C1=1 TSM_TUNE_CACHE=off
for x in 0 1 2 3 12 16 32 63; do
  if [ $x = 0 ]; then unset TSM_LIB_PATH; else export TSM_LIB_PATH=$R/workoutdetector_amd/libtsm_hip_x$x.so; fi
  echo -n "X=$x  "; timeout -k 10 200 python3 tools/c31_one.py 2>&1 | tail -1
done
