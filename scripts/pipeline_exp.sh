#!/bin/bash
# host batch-packing pipeline vs the reference-style structure around the same executor (csrc/host/pipeline_bench)
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
for mode in reference pipeline; do for prod in 4 8; do
  nshogi-engine_amd/csrc/host/pipeline_bench /tmp/w.nsgw $mode 6 $prod 512 2 4 4 | tail -1 | cut -c1-300
done; done
