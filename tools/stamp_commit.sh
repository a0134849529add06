#!/bin/bash
# Run HERE before a profiling gpurun: the GPU box has no .git, so the commit the tree derives from
# travels as a file (tools/summarize_profile.py stamps it into profiles/traffic.json; "+dirty" when
# the working tree differs from it).
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- tantivy_amd/csrc || c="$c+dirty"
echo "$c" > .git_commit_stamp
echo "$c"
