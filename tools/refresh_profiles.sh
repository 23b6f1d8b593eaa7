#!/usr/bin/env bash
# Re-measures everything profiles/ holds for one tag, on a 1-GPU MI355X box (run from the repo root, e.g. through
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02_a'
# ).  Writes under gpurun_out/<tag>/ and copies the summaries into profiles/<tag>_*.  PMC passes are separate
# rocprofv3 runs (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) and carry --kernel-trace only.  The PMC
# tables are stamped with the hash of the kernel sources (build.source_hash): bench.py accepts them for this tree only.
set -uo pipefail
TAG=${1:?usage: refresh_profiles.sh <tag>}
OUT=gpurun_out/$TAG
rm -rf "$OUT"        # a reused tag must not mix passes (gpurun merges the box's gpurun_out/ into the local one: clear that too)
mkdir -p "$OUT" profiles
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
QUIET="--no-cpu-baseline --no-pmc --no-extra-legs"

python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --steps 1 --warmup 1 $QUIET \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.err"; echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 1 --warmup 0 \
    --cells 131072 $QUIET > /dev/null 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 1 --warmup 0 \
    --cells 131072 $QUIET > /dev/null 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA \
    --output-format csv -d "$OUT/pmc_sq1" -- python3 bench.py --steps 1 --warmup 0 --cells 131072 $QUIET > /dev/null 2>&1; echo "sq1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY \
    --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py --steps 1 --warmup 0 --cells 131072 $QUIET > /dev/null 2>&1; echo "sq2 rc=$?"
python bench_train.py --steps 500 --warmup 30 > "$OUT/train_bench.json" 2> "$OUT/train_bench.err"; echo "train rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -- python3 bench_train.py --steps 200 --warmup 10 \
    --no-cpu-baseline > "$OUT/train_under_rocprof.json" 2> "$OUT/prof_train.err"; echo "train stats rc=$?"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_train" -- python3 tools/train_trace.py > "$OUT/train_trace.out" 2> /dev/null; echo "train trace rc=$?"
python tools/trace_gaps.py "$(ls -t "$OUT"/trace_train/*/*_kernel_trace.csv | head -1)" 200 > "$OUT/train_step_trace.txt"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_var" -- python3 tools/train_trace.py --variant > "$OUT/train_variant_trace.out" 2> /dev/null; echo "variant trace rc=$?"
python tools/trace_gaps.py "$(ls -t "$OUT"/trace_var/*/*_kernel_trace.csv | head -1)" 80 > "$OUT/train_variant_trace.txt"
python bench_train.py --variant --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/train_variant_bench.json" 2> /dev/null; echo "variant train rc=$?"
python bench_preprocess.py > "$OUT/preprocess_bench.json" 2> "$OUT/preprocess_bench.err"; echo "preprocess rc=$?"
python tools/bench_host_path.py > "$OUT/host_path.json" 2> "$OUT/host_path.err"; echo "host path rc=$?"
python tools/bench_variant.py > "$OUT/large_variant_bench.json" 2> "$OUT/large_variant_bench.err"; echo "variant rc=$?"
python tools/bench_fit.py --n 50000 > "$OUT/fit_bench.json" 2> "$OUT/fit_bench.err"; echo "fit rc=$?"
python tools/c12_diag.py > "$OUT/conv12_phase_diag.txt" 2> /dev/null; echo "diag rc=$?"
python tools/determinism_stress.py --reps 100 2> /dev/null | grep -v amdgpu.ids > "$OUT/determinism.txt"; echo "determinism rc=$?"

cp "$OUT/bench.json" "profiles/${TAG}_bench.json"
cp "$OUT/bench_under_rocprof.json" "profiles/${TAG}_bench_under_rocprof.json"
cp "$(ls -t "$OUT"/prof/*/*_kernel_stats.csv | head -1)" "profiles/${TAG}_kernel_stats.csv"
python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" 65536 > "profiles/${TAG}_pmc_traffic.json"
python tools/pmc_sq.py "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "profiles/${TAG}_sq_counters.json"
cp "$OUT/train_bench.json" "profiles/${TAG}_train_bench.json"
cp "$(ls -t "$OUT"/prof_train/*/*_kernel_stats.csv | head -1)" "profiles/${TAG}_train_kernel_stats.csv"
cp "$OUT/train_step_trace.txt" "profiles/${TAG}_train_step_trace.txt"
cp "$OUT/train_variant_trace.txt" "profiles/${TAG}_train_variant_trace.txt"
cp "$OUT/train_variant_bench.json" "profiles/${TAG}_train_variant_bench.json"
cp "$OUT/preprocess_bench.json" "profiles/${TAG}_preprocess_bench.json"
cp "$OUT/host_path.json" "profiles/${TAG}_host_path.json"
cp "$OUT/large_variant_bench.json" "profiles/${TAG}_large_variant_bench.json"
cp "$OUT/fit_bench.json" "profiles/${TAG}_fit_bench.json"
cp "$OUT/conv12_phase_diag.txt" "profiles/${TAG}_conv12_phase_diag.txt"
cp "$OUT/determinism.txt" "profiles/${TAG}_determinism.txt"
# the profiler's raw per-dispatch CSVs are large: keep the summaries only in what travels back
rm -rf "$OUT"/trace_train "$OUT"/trace_var "$OUT"/prof "$OUT"/prof_train "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_sq1 "$OUT"/pmc_sq2
mkdir -p "$OUT/profiles" && cp profiles/${TAG}_* "$OUT/profiles/"
echo "profiles/${TAG}_* refreshed (copies under $OUT/profiles/); add the rows to profiles/README.md"
