"""Which stream / hardware queue each kernel of a bench.py --force-gather run was dispatched on (rocprofv3 --kernel-trace csv):
    python tools/gather_stream.py <kernel_trace.csv>
The library issues the batch's gather (ncclSend / ncclRecv) on the pipeline's MATCHING stream, behind the batch's knn2 /
SearchForInitialization; the table shows the RCCL kernels on the same stream and queue as k_knn2_mfma / k_sfi_*."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cols = rows[0].keys() if rows else []
sid = next((c for c in ("Stream_Id", "Stream_ID", "stream_id") if c in cols), None)
qid = next((c for c in ("Queue_Id", "Queue_ID", "queue_id") if c in cols), None)
print("columns:", ", ".join(cols))
tab = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("orbfe::", "").replace("void ", "")[:48]
    tab[(r.get(sid, "?") if sid else "?", r.get(qid, "?") if qid else "?")][name] += 1
for (s, q), c in sorted(tab.items()):
    print("stream %s queue %s:" % (s, q))
    for name, n in c.most_common():
        print("    %6d  %s" % (n, name))
