#!/bin/bash
set -u
O=gpurun_out/r4j; mkdir -p $O
timeout 200 python tools/host_profile.py > $O/host_profile.txt 2>&1; echo rc=$?
head -70 $O/host_profile.txt | cut -c1-150
