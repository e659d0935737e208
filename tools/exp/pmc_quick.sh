R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="python $R/bench.py --steps 6 --warmup 0 --burn-in 2 --plies-per-step 256 --no-cpu-baseline --no-also"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_inst -o p -- $SHORT > $O/pmc_inst.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_act -o p -- $SHORT > $O/pmc_act.log 2>&1
find $O -name '*_agent_info.csv' -delete 2>/dev/null
python3 - <<PY
import csv,collections,glob
for sub in ('pmc_inst','pmc_act'):
    f=glob.glob('$O/%s/**/p_counter_collection.csv'%sub, recursive=True)
    rows=[r for r in csv.DictReader(open(f[0])) if 'k_rollout5' in r['Kernel_Name']]
    full=max(int(r['Grid_Size']) for r in rows)
    per=collections.defaultdict(dict)
    for r in rows:
        if int(r['Grid_Size'])==full: per[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
    ids=sorted(per)[2:]
    keys=sorted(per[ids[0]])
    for k in keys:
        v=sum(per[i][k] for i in ids)/len(ids)
        print(sub,k,'%.4g'%v,'per step %.3f'%(v/(65536*256)))
PY
