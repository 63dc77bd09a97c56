#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counters>" -- python3 script args...   (one pass, csv)
out=$1; shift; ctrs=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -- "$@" > $out.log 2>&1
