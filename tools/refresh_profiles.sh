#!/usr/bin/env bash
# Re-measures everything profiles/ holds for one tag, on a 1-GPU MI355X box (run from the repo root, e.g. through
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02_a'
# ).  Writes under gpurun_out/<tag>/ and copies the summaries into profiles/<tag>_*.  PMC passes are separate
# rocprofv3 runs (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) and carry --kernel-trace only.
set -euo pipefail
TAG=${1:?usage: refresh_profiles.sh <tag>}
OUT=gpurun_out/$TAG
rm -rf "$OUT"        # a reused tag must not mix passes (gpurun merges the box's gpurun_out/ into the local one: clear that too)
mkdir -p "$OUT" profiles
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null

python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 1 --warmup 0 \
    --cells 131072 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 1 --warmup 0 \
    --cells 131072 --no-cpu-baseline > /dev/null 2>&1
python bench_train.py --steps 500 --warmup 30 > "$OUT/train_bench.json" 2> "$OUT/train_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -- python3 bench_train.py --steps 200 --warmup 10 \
    --no-cpu-baseline > "$OUT/train_under_rocprof.json" 2> "$OUT/prof_train.err"
python bench_preprocess.py > "$OUT/preprocess_bench.json" 2> "$OUT/preprocess_bench.err"
python tools/bench_host_path.py > "$OUT/host_path.json" 2> "$OUT/host_path.err"
python tools/bench_variant.py > "$OUT/large_variant_bench.json" 2> "$OUT/large_variant_bench.err"
python tools/bench_fit.py --n 50000 > "$OUT/fit_bench.json" 2> "$OUT/fit_bench.err"

cp "$OUT/bench.json" "profiles/${TAG}_bench.json"
cp "$OUT/bench_under_rocprof.json" "profiles/${TAG}_bench_under_rocprof.json"
cp "$(ls -t "$OUT"/prof/*/*_kernel_stats.csv | head -1)" "profiles/${TAG}_kernel_stats.csv"
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" 65536 > "profiles/${TAG}_pmc_traffic.json"
cp "$OUT/train_bench.json" "profiles/${TAG}_train_bench.json"
cp "$(ls -t "$OUT"/prof_train/*/*_kernel_stats.csv | head -1)" "profiles/${TAG}_train_kernel_stats.csv"
cp "$OUT/preprocess_bench.json" "profiles/${TAG}_preprocess_bench.json"
cp "$OUT/host_path.json" "profiles/${TAG}_host_path.json"
cp "$OUT/large_variant_bench.json" "profiles/${TAG}_large_variant_bench.json"
cp "$OUT/fit_bench.json" "profiles/${TAG}_fit_bench.json"
echo "profiles/${TAG}_* refreshed; add the rows to profiles/README.md"
