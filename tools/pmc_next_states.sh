#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over gg_batch_next_states (tools/run_next_states.py), one counter group per run.
TAG=${1:-ns}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/run_next_states.py"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/inst -o p -- $CMD > $O/inst.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/act -o p -- $CMD > $O/act.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT --kernel-trace --output-format csv -d $O/lds -o p -- $CMD > $O/lds.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/mem -o p -- $CMD > $O/mem.log 2>&1
python - <<PY
import csv, glob, collections
for grp in ('inst','act','lds','mem'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % grp, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        for k, d in acc.items():
            if 'next_states' in k: print(grp, k, dict(d))
PY
