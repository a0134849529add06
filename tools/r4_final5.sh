#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python tools/vocab_sweep.py r04_vocab_sweep_b and2,mixed 256,4096,65536 2>&1 | grep -v amdgpu | tail -8
KEY_SUFFIX=_t4096 bash tools/profile_workload.sh and2 r04_and2_t4096 --terms 4096 > gpurun_out/prof_r04_and2_t4096.log 2>&1; tail -1 gpurun_out/prof_r04_and2_t4096.log | cut -c1-100
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_gpu_tests.log | tail -2
timeout 600 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_err.log; tail -c 150 gpurun_out/r04_bench_line.json
