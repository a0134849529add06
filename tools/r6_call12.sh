#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
VOCABS=256,4096,65536 bash tools/r5_stream.sh 2>&1 | tail -8
date
} > gpurun_out/r6_call12.txt 2>&1
