#!/bin/bash
# Round 5: pinned staging ring in DeviceLoader -- probe again (ring on / off), loader + padding tests
set -u
O=gpurun_out/r6s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/loader_stage_probe.py > $O/probe_ring.txt 2> $O/probe_ring.err; echo rc=$?
cat $O/probe_ring.txt
GPS_LOADER_PINNED_RING=0 timeout 600 python tools/loader_stage_probe.py > $O/probe_noring.txt 2> $O/probe_noring.err; echo rc=$?
cat $O/probe_noring.txt
timeout 900 python -m pytest tests/test_hip_optim.py tests/test_hip_padding.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
