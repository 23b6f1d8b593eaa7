#!/usr/bin/env bash
# Kernel-iteration loop on the GPU box: the conv parity tests, the phase diagnostic of the fused kernel, a bare bench.
set -uo pipefail
TAG=${1:?tag}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "fused_conv1 or each_layer or golden_cae or end_to_end or negative" > "$OUT/tests.log" 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error|p1 max err" "$OUT/tests.log" | tail -5
timeout -k 10 300 python tools/c12_diag.py > "$OUT/diag.log" 2>&1; echo "diag rc=$?"; cat "$OUT/diag.log" | tail -12
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], {k: (round(v["ms"] / j["steps"], 2), v.get("frac_executed")) for k, v in j["kernels"].items()})
PY
timeout -k 10 200 python tools/bench_variant.py > "$OUT/variant.json" 2> "$OUT/variant.err"; echo "variant rc=$?"; cat "$OUT/variant.json"
timeout -k 10 300 python -m pytest tests/test_gpu_large_variant.py -x -q -m gpu > "$OUT/tests_variant.log" 2>&1; echo "variant tests rc=$?"; tail -2 "$OUT/tests_variant.log"
