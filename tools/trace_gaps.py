"""A rocprofv3 kernel-trace CSV of tools/train_trace.py -> per training step (delimited by fit_gather_kernel): the span, the time
at least one kernel runs (union of the intervals of both streams), the idle remainder, launches, and the kernels by time.
    python tools/trace_gaps.py <..._kernel_trace.csv> [steps]"""
import collections
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
starts = [i for i, k in enumerate(ks) if "fit_gather_kernel" in k[2]][-(nsteps + 1):]
spans, busy = [], []
per = collections.defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    seg = ks[a:b]
    spans.append(ks[b][0] - seg[0][0])
    iv = sorted((s, e) for s, e, _, _ in seg)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy.append(tot + ce - cs)
    for s, e, n, q in seg:
        per[(n.replace("cs::(anonymous namespace)::", "").replace("cs::", "")[:90], q)].append(e - s)
print("steps %d  span %.1f us  busy (union of both streams) %.1f us  idle %.1f us  launches/step %.1f"
      % (len(spans), st.mean(spans) / 1e3, st.mean(busy) / 1e3, (st.mean(spans) - st.mean(busy)) / 1e3, sum(len(v) for v in per.values()) / len(spans)))
for (n, q), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print("%-92s q%s  n/step %.1f  mean %.2f us  per step %.1f us" % (n, q, len(v) / len(spans), st.mean(v) / 1e3, sum(v) / len(spans) / 1e3))
