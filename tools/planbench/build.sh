#!/bin/bash
# builds /tmp/plan_bench (host planner timing, no GPU): bash tools/planbench/build.sh
R=$(cd $(dirname $0)/../.. && pwd)
python -c "import sys; sys.path.insert(0,'$R'); from tantivy_amd import build as B; B.build()" >/dev/null 2>&1
OBJS=$(ls $R/tantivy_amd/lib/obj/*.hip.o $R/tantivy_amd/lib/obj/tq_*.cpp.o)
/opt/rocm/bin/hipcc -O3 -std=c++17 -Wno-unused-function -c $R/tools/planbench/plan_bench.cpp -o /tmp/plan_bench.o 2>/dev/null && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o /tmp/plan_bench /tmp/plan_bench.o $OBJS -ldl -lpthread
