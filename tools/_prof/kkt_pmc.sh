export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/kk_s1 $R/gpurun_out/kk_s2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/kk_s1 -o s1 -- python $R/tools/kkt_time.py 128 1024 > /dev/null 2> $R/gpurun_out/kk1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/kk_s2 -o s2 -- python $R/tools/kkt_time.py 128 1024 > /dev/null 2> $R/gpurun_out/kk2.err
cd $R
python tools/rocprof_summary.py pmc $(find gpurun_out/kk_s1 gpurun_out/kk_s2 -name "*.db") > gpurun_out/kk_pmc.json
rm -rf gpurun_out/kk_s1 gpurun_out/kk_s2
python - <<'PY'
import json
d=json.load(open('gpurun_out/kk_pmc.json'))
for k,v in d.items():
    if 'kkt' in k: print(k, json.dumps(v, indent=0)[:1500])
PY

