#!/bin/bash
# ktrace.sh <tag> <command...>: rocprofv3 kernel trace + stats of a command into gpurun_out/<tag>/ (run on the GPU box).
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG -o t -- "$@" > $ROOT/gpurun_out/$TAG.log 2>&1
