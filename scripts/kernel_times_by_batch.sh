#!/bin/bash
# per-kernel average times of the bench workload at several batch sizes: scripts/kernel_times_by_batch.sh 1 16 64
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in "$@"; do
  echo "== batch $b"
  rm -rf gpurun_out/kt_tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tmp -- python3 bench.py --workload-only 40 --batch $b > gpurun_out/kt_tmp.log 2>&1
  python3 scripts/kernel_stats_top.py gpurun_out/kt_tmp 8
done
