#!/bin/bash
set -u
OUT=gpurun_out/r6_13; mkdir -p $OUT
export TMPDIR=/tmp
X=$PWD/tools/micro/exp/libgps_hip_exp.so
for cfg in "default::" "s3:$X:" "s3nj2mb1:$X:GPS_EXP_NJ=2,GPS_EXP_MB=1" "s3nj2mb2:$X:GPS_EXP_NJ=2,GPS_EXP_MB=2" "s3nj3mb1:$X:GPS_EXP_MB=1"; do
  name=${cfg%%:*}; rest=${cfg#*:}; lib=${rest%%:*}; envs=${rest#*:}
  ( [ -n "$lib" ] && export GPS_HIP_LIB=$lib; for kv in ${envs//,/ }; do export "$kv"; done
    rm -rf /tmp/p_$name; rocprofv3 --kernel-trace -d /tmp/p_$name -o t -- python tools/gemm_panel_bench.py > $OUT/$name.txt 2>&1
    python tools/rocpd_stats.py $(find /tmp/p_$name -name "*.db" | head -1) --match "ring16" --top 12 > $OUT/${name}_kern.txt )
  echo "== $name"; grep -E "^pq|^out|^C  |^ff1|^ff2|^dgrad g" $OUT/$name.txt | cut -c1-30,62-100; cut -c1-150 $OUT/${name}_kern.txt | sed -n 2,12p
done
