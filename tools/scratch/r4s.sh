mkdir -p gpurun_out/r4s
python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -x -q > gpurun_out/r4s/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4s/tests.log; tail -4 gpurun_out/r4s/tests.log
python bench.py --no-cpu-baseline --no-pmc --no-extra-legs > gpurun_out/r4s/bench.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/r4s/bench.json'));print(d['value'],d['ms_per_step'],{k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r4s/pmc -- python3 bench.py --steps 1 --warmup 0 --cells 131072 --no-cpu-baseline --no-pmc --no-extra-legs > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/r4s/pmc/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "conv" in k: print(k, {a:int(b) for a,b in v.items()})
PY
rm -rf gpurun_out/r4s/pmc
