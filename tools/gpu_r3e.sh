#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B=${1:-tools/rdbt_check}
(cd /tmp && rocprofv3 --pmc ${PMC:-SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY} \
   --kernel-trace --output-format csv -d /tmp/pmc_r3e -- $R/$B time32 > /tmp/pmc_r3e.log 2>&1)
python tools/pmc_sq.py /tmp/pmc_r3e $O/r03e_pmc_icache.json > /dev/null || tail -20 /tmp/pmc_r3e.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03e_pmc_icache.json"))
for k,v in d.items(): print(k, {a:round(b) for a,b in v["per_launch"].items()})
PY
