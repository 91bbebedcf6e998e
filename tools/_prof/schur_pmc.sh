# SQ counter passes of the Schur formation kernels (tools/time_schur.py 128 1024) under rocprofv3; run on the GPU box.
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/sc_s1 $R/gpurun_out/sc_s2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $R/gpurun_out/sc_s1 -o s1 -- python $R/tools/time_schur.py 128 1024 > /dev/null 2> $R/gpurun_out/sc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $R/gpurun_out/sc_s2 -o s2 -- python $R/tools/time_schur.py 128 1024 > /dev/null 2> $R/gpurun_out/sc2.err
cd $R
python tools/rocprof_summary.py pmc $(find gpurun_out/sc_s1 gpurun_out/sc_s2 -name "*.db") > gpurun_out/sc_pmc.json
rm -rf gpurun_out/sc_s1 gpurun_out/sc_s2
python - <<'PY'
import json
d=json.load(open('gpurun_out/sc_pmc.json'))['kernels']
for k,v in d.items():
    if 'schur' in k or 'complete_ss' in k or 'invert_g' in k:
        print(k)
        for c,x in v.items(): print('   %-22s %14.0f  (%.1f us, %d records)' % (c, x['avg'], x.get('avg_duration_us_profiled',0), x['launches']))
PY
