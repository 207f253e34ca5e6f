#!/bin/bash
# c2 (headline) time budget: GPU idle between kernels over the timed region (kernel trace)
set -u
O=gpurun_out/${1:-r06_r}; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o c2 --output-format csv -- python bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_events > $O/r06_r_c2_prof.json 2> $O/r06_r_c2_prof.err
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" > $O/r06_r_c2_gaps.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t1 = int(rows[-1]["End_Timestamp"])
cut = t1 - 6 * 62.5e6
sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
# concurrent streams: merge intervals for the busy time
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
per = collections.defaultdict(lambda: [0, 0]); gapafter = collections.defaultdict(lambda: [0, 0])
for a, b in zip(sel, sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"]); k = a["Kernel_Name"][:70]
    gapafter[k][0] += max(g, 0); gapafter[k][1] += 1
for r in sel:
    k = r["Kernel_Name"][:70]; per[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[k][1] += 1
print("kernel, launches, total ms, avg us, idle-after total ms, avg idle-after us")
for k, (tt, n) in sorted(per.items(), key=lambda kv: -gapafter[kv[0]][0])[:40]:
    ga = gapafter[k]
    print(f"{k:70s} {n:6d} {tt/1e6:8.2f} {tt/n/1e3:8.1f} {ga[0]/1e6:8.2f} {ga[0]/max(ga[1],1)/1e3:8.1f}")
PY
rm -rf $O/prof
head -30 $O/r06_r_c2_gaps.txt
