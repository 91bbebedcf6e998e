# SQ / memory counter passes of the round-4 producer kernels (tools/_prof/run_walk.py) under rocprofv3; run on the GPU box.
export TMPDIR=/tmp; R=$PWD; cd /tmp
for d in wk_s1 wk_s2 wk_f wk_w; do rm -rf $R/gpurun_out/$d; done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $R/gpurun_out/wk_s1 -o s1 -- python $R/tools/_prof/run_walk.py $1 $2 > /dev/null 2> $R/gpurun_out/wk1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS -d $R/gpurun_out/wk_s2 -o s2 -- python $R/tools/_prof/run_walk.py $1 $2 > /dev/null 2> $R/gpurun_out/wk2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/wk_f -o f -- python $R/tools/_prof/run_walk.py $1 $2 > /dev/null 2> $R/gpurun_out/wk3.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/wk_w -o w -- python $R/tools/_prof/run_walk.py $1 $2 > /dev/null 2> $R/gpurun_out/wk4.err
cd $R
python tools/rocprof_summary.py pmc $(find gpurun_out/wk_s1 gpurun_out/wk_s2 gpurun_out/wk_f gpurun_out/wk_w -name "*.db") > gpurun_out/wk_pmc.json
for d in wk_s1 wk_s2 wk_f wk_w; do rm -rf gpurun_out/$d; done
python - <<'PY'
import json
d=json.load(open('gpurun_out/wk_pmc.json'))['kernels']
for k,v in d.items():
    if 'schur' in k or 'dz' in k:
        print(k)
        for c,x in v.items(): print('   %-22s %14.0f  (%.1f us, %d records)' % (c, x['avg'], x.get('avg_duration_us_profiled',0), x['launches']))
PY
tail -n 2 gpurun_out/wk*.err
