#!/bin/bash
# PMC counters of the three arithmetic builds of generate_kkt_kernel (run on the GPU box): two rocprofv3 --pmc passes over tools/_prof/kkt_f32.py -> gpurun_out/<tag>_kkt_pmc.json
TAG=${1:-r06f}
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/kp1 $R/gpurun_out/kp2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $R/gpurun_out/kp1 -o a -- python $R/tools/_prof/kkt_f32.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d $R/gpurun_out/kp2 -o b -- python $R/tools/_prof/kkt_f32.py > /dev/null 2>&1
cd $R
python tools/rocprof_summary.py pmc $(find gpurun_out/kp1 gpurun_out/kp2 -name "*.db") > gpurun_out/${TAG}_kkt_pmc.json
rm -rf gpurun_out/kp1 gpurun_out/kp2
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_kkt_pmc.json"))["kernels"]
for k, v in d.items():
    if "generate_kkt" in k:
        g = lambda c: v.get(c, {}).get("avg", 0)
        print(k[:72], "| VALU pipe busy", round(4 * g("SQ_ACTIVE_INST_VALU") / 32 / max(g("SQ_BUSY_CYCLES"), 1), 3), "| VALU instructions", int(g("SQ_INSTS_VALU")),
              "| busy cycles", int(g("SQ_BUSY_CYCLES")), "| LDS conflict share", round(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), 3),
              "| LDS active per CU / busy", round(g("SQ_LDS_IDX_ACTIVE") / 8 / max(g("SQ_BUSY_CYCLES"), 1), 3))
PY
