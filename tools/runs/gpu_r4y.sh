#!/bin/bash
# final tree: the default bench line
set -u
O=gpurun_out/r4y; mkdir -p $O
timeout 170 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_final.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value']), d['launch_mode'], 'host', round(d['host_enqueue_ms_per_step'],2), 'pcie', d.get('pcie_inclusive_ms_per_step'), d['launch_trial_ms'], d['roofline']['frac'], d['roofline']['traffic_source'], d['cpu_baseline']['value'])"
