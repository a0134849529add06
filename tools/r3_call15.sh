#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
BENCH_SIDE_ORDER=or5 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c15_kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --latency-queries 0 > $R/gpurun_out/c15.log 2>&1
python - <<'PY'
import csv,glob,os,json
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
f=glob.glob(R+'/gpurun_out/c15_kt/**/kt_kernel_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows=[r for r in rows if 'ushare_kernel<2>' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print([round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3) for r in rows])
l=open(R+'/gpurun_out/c15.log').read().strip().splitlines()
j=json.loads([x for x in l if x.startswith('{')][-1])
print('bench says', j['other_workloads']['or5']['kernel_ms_avg'])
PY
