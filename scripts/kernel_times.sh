#!/bin/bash
# per-kernel average times of one bench run (rocprofv3 --kernel-trace --stats): scripts/kernel_times.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tmp -- python3 bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --steps 6 --warmup 2 "$@" > gpurun_out/kt_tmp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt_tmp/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4s}  {r['Percentage']:>6s}%  {r['Name'][:90]}")
PY
