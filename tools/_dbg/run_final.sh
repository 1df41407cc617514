set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_mfma_r02 gpurun_out/prof_final
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_bench/f -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> gpurun_out/pmc_f.err
echo "fetch pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_bench/w -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> gpurun_out/pmc_w.err
echo "write pass done"
python tools/pmc_bench_traffic.py gpurun_out/pmc_bench/f gpurun_out/pmc_bench/w gpurun_out/igemm_traffic.json
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-include-regex "igemm_kernel|attn_kernel" -d gpurun_out/pmc_mfma_r02/p -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> gpurun_out/pmc_mfma.err
python tools/pmc_mfma_util.py gpurun_out/pmc_mfma_r02/p gpurun_out/r02_mfma_utilisation.json
echo "mfma pass done"
rm -rf gpurun_out/pmc_bench gpurun_out/pmc_mfma_r02
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o p --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/prof.err
cp $(ls gpurun_out/prof_final/*/*kernel_stats.csv gpurun_out/prof_final/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/rocprofv3_kernel_stats_final.csv
rm -rf gpurun_out/prof_final
echo "stats done"
cp gpurun_out/igemm_traffic.json profiles/igemm_traffic.json
timeout -k 10 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 600 gpurun_out/bench_final.json
