# Round profile recipe (run on the GPU box through gpurun):  bash tools/profile_round.sh r04x
#   1. GPU tests   2. un-profiled bench line (the driver's command: --steps 20 --warmup 5) + the full record   3. rocprofv3 --kernel-trace --stats of the same bench command
#   4. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets) of `bench.py --profile-lean`   5. summaries -> gpurun_out/<tag>_*
# Copy gpurun_out/<tag>_* into profiles/ and run tools/make_traffic.py afterwards (here, not on the box).
TAG=${1:-r04}
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/${TAG}_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cp gpurun_out/bench_full.json gpurun_out/${TAG}_bench_full.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
for d in prof pmc_f pmc_w pmc_s1 pmc_s2; do rm -rf $R/gpurun_out/${TAG}_$d; done
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --full-json '' > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/rocprof1.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_f -o f -- python $R/bench.py --steps 3 --warmup 1 --profile-lean --full-json '' > /dev/null 2> $R/gpurun_out/rocprof2.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc_w -o w -- python $R/bench.py --steps 3 --warmup 1 --profile-lean --full-json '' > /dev/null 2> $R/gpurun_out/rocprof3.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/${TAG}_pmc_s1 -o s1 -- python $R/bench.py --steps 3 --warmup 1 --profile-lean --full-json '' > /dev/null 2> $R/gpurun_out/rocprof4.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/${TAG}_pmc_s2 -o s2 -- python $R/bench.py --steps 3 --warmup 1 --profile-lean --full-json '' > /dev/null 2> $R/gpurun_out/rocprof5.err
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/${TAG}_prof -name "*.db" | head -1) > gpurun_out/${TAG}_kernel_stats.txt
python tools/rocprof_summary.py pmc $(find gpurun_out/${TAG}_pmc_f gpurun_out/${TAG}_pmc_w gpurun_out/${TAG}_pmc_s1 gpurun_out/${TAG}_pmc_s2 -name "*.db") > gpurun_out/${TAG}_pmc.json
for d in prof pmc_f pmc_w pmc_s1 pmc_s2; do rm -rf gpurun_out/${TAG}_$d; done
cat gpurun_out/${TAG}_pytest_gpu.log; head -14 gpurun_out/${TAG}_kernel_stats.txt; tail -c 300 gpurun_out/${TAG}_bench.json; for f in gpurun_out/rocprof*.err; do tail -n 3 $f; done
