#!/bin/bash
set -u
O=gpurun_out/r2i; mkdir -p $O
bash tools/pmc_collect.sh $O > $O/collect.log 2>&1
cat $O/pmc_3.txt $O/pmc_4.txt 2>/dev/null | grep -E "sattn|gatedgcn" | cut -c1-170
ls $O
