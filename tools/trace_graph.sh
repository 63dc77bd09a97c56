#!/bin/bash
# rocprofv3 kernel trace of tools/try_graph.py (graph replays of the config-3 step) -> gpurun_out/trace_<tag>/
# usage: tools/trace_graph.sh <tag>   (environment switches are inherited)
tag=$1
root=$GRAFT_REPO_ROOT
[ -z "$root" ] && root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/tools/try_graph.py c3 > $out/log.txt 2>&1
grep "ms/step" $out/log.txt
