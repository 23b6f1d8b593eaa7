"""Development: from a rocprofv3 kernel-trace CSV, the last 200 training steps: span per step, busy time (union of kernel intervals), per-kernel mean."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ks.sort()
# steps delimited by fit_gather_kernel
starts = [i for i, k in enumerate(ks) if "fit_gather_kernel" in k[2]]
starts = starts[-81:]
spans, busy = [], []
per = collections.defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    seg = ks[a:b]
    spans.append(seg[-1][1] - seg[0][0] if False else ks[b][0] - seg[0][0])
    iv = sorted((s, e) for s, e, _, _ in seg)
    tot, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    busy.append(tot)
    for s, e, n, q in seg: per[(n.replace("cs::(anonymous namespace)::","").replace("cs::","")[:90], q)].append(e - s)
import statistics as st
print("steps", len(spans), "span us %.1f" % (st.mean(spans) / 1e3), "busy (union) us %.1f" % (st.mean(busy) / 1e3), "idle us %.1f" % ((st.mean(spans) - st.mean(busy)) / 1e3), "launches/step %.1f" % (sum(len(v) for v in per.values()) / len(spans)))
for (n, q), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("%-92s q%s  n/step %.1f  mean us %.2f  total/step us %.1f" % (n, q, len(v) / len(spans), st.mean(v) / 1e3, sum(v) / len(spans) / 1e3))
