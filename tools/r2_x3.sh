#!/usr/bin/env bash
# split-bf16 A/B on the GPU box: parity tests of the reference graph and of the 128x128 variant, then benches with and without it.
set -uo pipefail
TAG=${1:?tag}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s  > "$OUT/tests.log" 2>&1; echo "tests rc=$?"
grep -E "passed|failed|error|conv4 max" "$OUT/tests.log" | tail -5
timeout -k 10 600 python -m pytest tests/test_gpu_large_variant.py -x -q -m gpu -s > "$OUT/tests_variant.log" 2>&1; echo "variant tests rc=$?"; tail -3 "$OUT/tests_variant.log"
for v in on off; do
  if [ $v = off ]; then export CS_NO_BF16X3=1; fi
  timeout -k 10 200 python tools/bench_variant.py > "$OUT/variant_$v.json" 2> "$OUT/variant_$v.err"; echo "variant $v rc=$?"; cat "$OUT/variant_$v.json"
done
unset CS_NO_BF16X3
for v in on noconv6 off; do
  unset CS_NO_BF16X3 CS_NO_BF16X3_CONV6
  if [ $v = off ]; then export CS_NO_BF16X3=1; fi
  if [ $v = noconv6 ]; then export CS_NO_BF16X3_CONV6=1; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"; echo "bench $v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], {k: (round(v["ms"] / j["steps"], 2), v.get("frac_executed"), v.get("frac_bf16_mfma_peak")) for k, v in j["kernels"].items()})
PY
done
unset CS_NO_BF16X3 CS_NO_BF16X3_CONV6
CS_WINO_DIAG=1 timeout -k 10 200 python tools/wino_diag.py > "$OUT/wino_diag.log" 2>&1; echo "diag rc=$?"; tail -8 "$OUT/wino_diag.log"
