#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
VOCABS=65536,1048576 bash tools/r5_stream.sh 2>&1 | tail -4

date
} > gpurun_out/r6_call14.txt 2>&1
