#!/bin/bash
# headline rate against the chunk size (cells per launch)
for c in 65536 131072 262144 65536; do
  timeout -k 10 200 python bench.py --chunk $c --no-pmc --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['config']['chunk_cells'], j['value'], j['ms_per_step'])"
done
