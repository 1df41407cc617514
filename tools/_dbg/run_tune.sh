set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python tools/autotune_insitu.py --batches 4,8,1,2,16 --merge pbe_amd/tuned_mi355x.json --out gpurun_out/tuned_insitu.json --report gpurun_out/autotune_insitu_report.txt > gpurun_out/tune.log 2>&1
tail -3 gpurun_out/tune.log
