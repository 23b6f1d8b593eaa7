#!/usr/bin/env bash
# fused conv6 + conv7 kernel iteration: parity tests that cross it, the bare bench, phase stamps.
set -uo pipefail
TAG=${1:?tag}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -2 "$OUT/tests.log"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], {k: (round(v["ms"] / j["steps"], 2), v.get("frac_executed"), v.get("frac_bf16_mfma_peak")) for k, v in j["kernels"].items()})
PY
CS_WINO_DIAG=1 timeout -k 10 200 python tools/wino_diag.py > "$OUT/wino_diag.log" 2>&1; echo "diag rc=$?"; tail -8 "$OUT/wino_diag.log"
