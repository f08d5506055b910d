#!/bin/bash
# SQ counters of the dense-block kernels from the stand-alone harness (time mode: old, 8x8, 8x16; fwd + bwd; N = 32/16/64)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY \
   --kernel-trace --output-format csv -d /tmp/pmc_r3d -- $R/tools/rdbt_check time > /tmp/pmc_r3d.log 2>&1)
python tools/pmc_sq.py /tmp/pmc_r3d $O/r03d_pmc_sq.json || tail -20 /tmp/pmc_r3d.log
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM \
   --kernel-trace --output-format csv -d /tmp/pmc_r3d2 -- $R/tools/rdbt_check time > /tmp/pmc_r3d2.log 2>&1)
python tools/pmc_sq.py /tmp/pmc_r3d2 $O/r03d_pmc_sq2.json || tail -20 /tmp/pmc_r3d2.log
python - <<'PY'
import json
for f in ("gpurun_out/r03d_pmc_sq.json","gpurun_out/r03d_pmc_sq2.json"):
    try:
        d=json.load(open(f))
        for k,v in d.items(): print(k, {a:round(b) for a,b in v["per_launch"].items()})
    except Exception as e: print(f, e)
PY
