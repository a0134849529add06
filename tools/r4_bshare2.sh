#!/bin/bash
# waves-per-SIMD variants of the boolean shared launch
export PROBE_VOCAB=24
for v in "" bs5 bs4; do
  if [ -n "$v" ]; then export TQ_LIB_PATH=$PWD/tantivy_amd/lib/variants/libtantivy_amd_$v.so; fi
  echo "variant ${v:-base}"
  python tools/probe_bshare.py
  for sh in 1 2; do PROBE_SHAPE=$sh python tools/probe_bshare.py; done
done
