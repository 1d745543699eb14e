#!/bin/bash
# gpurun wrapper: stamps the tree's identity into .head_sha (shipped with the snapshot, not tracked) so that what runs on the GPU box can
# name the commit it ran.  usage: tools/gpu.sh TIMEOUT 'command'
cd "$(dirname "$0")/.."
sha=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- . ':!gpurun_out' || sha="${sha}-dirty"
echo "$sha" > .head_sha
python -m open_clip_amd.build > /dev/null && python -m open_clip_amd.build --dev > /dev/null && python -m pytest tests/test_cabi.py -q -x > /dev/null || { echo "build / C-ABI check failed: not calling the GPU box"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
