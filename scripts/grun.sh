#!/bin/bash
# build() here (hipcc cross-compiles), then run a command on the GPU box: bash scripts/grun.sh <timeout> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
