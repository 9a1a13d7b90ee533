#!/bin/bash
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
mkdir -p gpurun_out/r2
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
rm -rf gpurun_out/kt_sp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_sp -- nshogi-engine_amd/csrc/selfplay/selfplay --executor hip --weights /tmp/w.nsgw --gpu 0 --threads 1 --workers 8 --solver-threads 4 --games-per-group 128 --playouts 800 --seconds 20 --seed 1 --precision 5 > gpurun_out/r2/sp_trace.json 2> gpurun_out/r2/sp_trace.err
python3 - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/kt_sp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
d = json.loads(open('gpurun_out/r2/sp_trace.json').read().strip().splitlines()[-1])
print("kernel time total %.3f s of %.1f s wall = %.3f busy; batches %d" % (tot/1e9, d['seconds'], tot/1e9/d['seconds'], round(d['evals_per_sec']*d['seconds']/d['avg_batch'])))
print({k: d[k] for k in ('evals_per_sec','avg_batch','await_ms_per_batch','host_ms_per_batch')})
for r in rows[:6]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>7s}  {r['Percentage']:>6s}%  {r['Name'][:70]}")
PY
rm -rf gpurun_out/kt_sp
