set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
timeout 400 python bench.py --spmv --latency > gpurun_out/r01k_bench.json 2> gpurun_out/r01k_bench.err
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/prof_r01k $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01k -o r01k -- python $R/bench.py --steps 8 --warmup 1 --spmv --no-cpu-baseline > $R/gpurun_out/r01k_bench_under_rocprof.json 2> $R/gpurun_out/rocprof1.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f -o f -- python $R/bench.py --steps 2 --warmup 1 --spmv --no-cpu-baseline > /dev/null 2> $R/gpurun_out/rocprof2.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w -o w -- python $R/bench.py --steps 2 --warmup 1 --spmv --no-cpu-baseline > /dev/null 2> $R/gpurun_out/rocprof3.err
cd $R
find gpurun_out/prof_r01k gpurun_out/pmc_f gpurun_out/pmc_w -name "*.db" | head
python tools/rocprof_summary.py stats $(find gpurun_out/prof_r01k -name "*.db" | head -1) > gpurun_out/r01k_kernel_stats.txt
python tools/rocprof_summary.py pmc $(find gpurun_out/pmc_f -name "*.db" | head -1) $(find gpurun_out/pmc_w -name "*.db" | head -1) > gpurun_out/r01k_pmc.json
rm -rf gpurun_out/prof_r01k gpurun_out/pmc_f gpurun_out/pmc_w
cat gpurun_out/pytest_gpu.log; head -12 gpurun_out/r01k_kernel_stats.txt; tail -c 400 gpurun_out/r01k_bench.json
