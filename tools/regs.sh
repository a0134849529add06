#!/bin/bash
# static register / scratch / LDS figures of one kernel file under extra -D flags (no GPU needed):
#   bash tools/regs.sh tq_ushare.hip [-DTQ_US_SPEC=1 ...]
U=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -Rpass-analysis=kernel-resource-usage \
  -c $R/tantivy_amd/csrc/$U -o /tmp/regs/$U.$$.o 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:" | \
  sed -e 's/.*remark: [^ ]* *//' | paste - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
rm -f /tmp/regs/$U.$$.o
