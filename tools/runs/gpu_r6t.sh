#!/bin/bash
set -u
O=gpurun_out/r6t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
TRACE_ONE_THREAD=1 timeout 600 python tools/loader_stage_trace.py > $O/trace.txt 2> $O/trace.err; echo rc=$?
cat $O/trace.txt; tail -3 $O/trace.err
