#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
export TQ_US_TASK_COST=512
bash tools/profile_workload.sh or5 r03a_or5 > gpurun_out/c4_prof.log 2>&1
tail -5 gpurun_out/c4_prof.log
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4_kt -o kt -- python $R/bench.py --workload or5 --steps 2 --warmup 1 --no-cpu-baseline --latency-queries 0 --no-side > $R/gpurun_out/c4_kt.log 2>&1
python - <<'PY'
import csv,glob,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
f=glob.glob(R+'/gpurun_out/c4_kt/**/kt_kernel_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows=[r for r in rows if 'ushare' in r['Kernel_Name'] or 'merge_lists' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-14:]:
    print(r['Kernel_Name'][:50], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us', 'grid',r.get('Grid_Size_X', r.get('Grid_Size')))
PY
