#!/bin/bash
set -u
O=gpurun_out/r2b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
tail -15 $O/pytest.log
timeout 900 python tools/gg_sweep.py > $O/gg_sweep.txt 2>&1; echo "sweep rc=$?" >> $O/rc.txt
cat $O/gg_sweep.txt
cat $O/rc.txt
