#!/bin/bash
# c3 (host envs, async): GPU idle / per-stream busy over the timed region
set -u
O=gpurun_out/${1:-r06_y}; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
python bench.py --workload c3 --steps 12 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_events 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('c3 plain', d['ms_per_step'], d['value'], json.dumps(d.get('ingest')))" | tee $O/r06_y_c3.log
rocprofv3 --kernel-trace -d $O/prof -o c3 --output-format csv -- python bench.py --workload c3 --steps 12 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_events > $O/r06_y_c3_prof.json 2> $O/r06_y_c3_prof.err
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" > $O/r06_y_c3_gaps.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t1 = int(rows[-1]["End_Timestamp"])
cut = t1 - 12 * 62e6
sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = iv[-1][1] - iv[0][0]
print(f"kernels {len(sel)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms  ({100*(span-busy)/span:.2f} %)")
qs = collections.defaultdict(lambda: [0, 0])
for r in sel:
    q = r.get("Queue_Id", "?"); qs[q][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); qs[q][1] += 1
for q, (tt, n) in sorted(qs.items(), key=lambda kv: -kv[1][0]):
    print(f"queue {q}: {n} kernels, {tt/1e6:.2f} ms")
per = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"][:70]; per[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[k][1] += 1
for k, (tt, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"{k:70s} {n:6d} {tt/1e6/12:8.2f} ms/step {tt/n/1e3:8.1f} us")
PY
rm -rf $O/prof
head -40 $O/r06_y_c3_gaps.txt
