#!/usr/bin/env bash
# One GPU-box call of the development loop: GPU tests, the default bench (with its own PMC passes), the same under
# rocprofv3 --stats, and an A/B leg with the previous two-kernel conv1 / conv2 path.  Output under gpurun_out/<tag>/.
set -uo pipefail
TAG=${1:?usage: r2_quick.sh <tag> [pytest -k expr]}
KEXPR=${2:-}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$KEXPR" -s > "$OUT/tests.log" 2>&1; echo "tests rc=$?" | tee -a "$OUT/tests.log"
else
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1; echo "tests rc=$?" | tee -a "$OUT/tests.log"
fi
tail -5 "$OUT/tests.log"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-extra-legs \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.err"; echo "prof rc=$?"
cp "$(ls -t "$OUT"/prof/*/*_kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/prof"
CS_NO_FUSE12=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > "$OUT/bench_nofuse12.json" 2> "$OUT/bench_nofuse12.err"; echo "ab rc=$?"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for f in ("bench.json", "bench_nofuse12.json"):
    try:
        j = json.loads(open(out + "/" + f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], {k: (v["ms"], v.get("frac_executed")) for k, v in j["kernels"].items()})
        print("  roofline", {k: j["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "traffic_source")})
        for k in ("cpu_baseline", "parity_on_cpu_sample", "parity_vs_reference_sequence", "small_n", "train_leg"):
            if k in j: print("  ", k, json.dumps(j[k])[:600])
    except Exception as e:
        print(f, "unreadable:", e)
PY
timeout -k 10 200 python tools/bench_variant.py > "$OUT/large_variant_bench.json" 2> "$OUT/large_variant_bench.err"; echo "variant rc=$?"; cat "$OUT/large_variant_bench.json"
timeout -k 10 200 python bench_train.py --steps 500 --warmup 30 > "$OUT/train_bench.json" 2> "$OUT/train_bench.err"; echo "train rc=$?"; cat "$OUT/train_bench.json" | cut -c1-400
