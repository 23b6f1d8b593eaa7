#!/usr/bin/env bash
# conv4 split-bf16 A/B on the GPU box: parity tests that cross conv4, then the bare bench with and without it.
set -uo pipefail
TAG=${1:?tag}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "each_layer or golden_cae or end_to_end or negative or stage or conv4" > "$OUT/tests.log" 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error|err" "$OUT/tests.log" | tail -15
for v in on off; do
  if [ $v = off ]; then export CS_NO_BF16X3=1; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"; echo "bench $v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], {k: (round(v["ms"] / j["steps"], 2), v.get("frac_executed")) for k, v in j["kernels"].items()})
PY
done
