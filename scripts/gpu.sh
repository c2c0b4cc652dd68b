#!/bin/bash
# build + load check here, then run the given command on the MI355X box:  scripts/gpu.sh [timeout_s] 'command'
set -e
cd "$(dirname "$0")/.."
T=900
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
python -m graphvqa_amd.build | tail -1
python -c "from graphvqa_amd import _lib; _lib.load()"
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
