#!/bin/bash
# round 4, last evidence call: smoke, GPU tests, boolean profile, vocabulary sweep, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_gpu_tests.log | tail -2
bash tools/profile_workload.sh bool r04_bool > gpurun_out/prof_r04_bool.log 2>&1; tail -1 gpurun_out/prof_r04_bool.log | cut -c1-100
timeout 900 python tools/vocab_sweep.py r04_vocab_sweep 2>&1 | grep -v amdgpu | tail -17
timeout 600 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_err.log; tail -c 600 gpurun_out/r04_bench_line.json
