set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/diag5_stamps.txt
: > $O
PBE_STAMPS_LIB=$GRAFT_REPO_ROOT/tools/_dbg/libpbe_hip_lat_stamps.so PBE_STAMP_CFG=10 timeout -k 10 300 python tools/phase_stamps.py c:8:64:64:320:0:320 c:8:16:16:1280:0:1280 >> $O 2>&1
tail -4 $O
