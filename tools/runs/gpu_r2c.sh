#!/bin/bash
set -u
O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "gatedgcn or graph_index" > $O/pytest_gg.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -3 $O/pytest_gg.log
timeout 900 python tools/gg_sweep.py > $O/gg_sweep.txt 2>&1; echo "sweep rc=$?" >> $O/rc.txt
cat $O/gg_sweep.txt
cat $O/rc.txt
