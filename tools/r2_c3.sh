#!/usr/bin/env bash
# conv3 Winograd split-bf16 A/B: the parity file, then the bare bench both ways.
set -uo pipefail
TAG=${1:?tag}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -3 "$OUT/tests.log"
for v in on off; do
  unset CS_NO_BF16X3_CONV3
  if [ $v = off ]; then export CS_NO_BF16X3_CONV3=1; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"; echo "bench $v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], {k: (round(v["ms"] / j["steps"], 2), v.get("frac_matrix_peaks")) for k, v in j["kernels"].items()})
PY
done
