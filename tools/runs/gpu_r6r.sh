#!/bin/bash
# Round 5: where the staging thread's time goes when it pads (tools/loader_stage_probe.py)
set -u
O=gpurun_out/r6r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/loader_stage_probe.py > $O/probe.txt 2> $O/probe.err; echo rc=$?
cat $O/probe.txt; tail -5 $O/probe.err
