#!/bin/bash
# c5 (configs[4]) time budget: kernel-trace stats of the c5 bench (sum of kernel time vs wall ms/step)
set -u
O=gpurun_out/${1:-r06_p}; mkdir -p $O
cd /root/repo
export TMPDIR=/tmp
python bench.py --workload c5 --steps 16 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_events > $O/r06_p_c5_plain.json 2> $O/r06_p_c5_plain.err
rocprofv3 --kernel-trace --stats -d $O/prof -o c5 --output-format csv -- python bench.py --workload c5 --steps 16 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_events > $O/r06_p_c5_prof.json 2> $O/r06_p_c5_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/r06_p_c5_kernel_stats.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" > $O/r06_p_c5_gaps.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 16 steps' worth: take the final 60 % of the trace by time
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t1 - 16 * 16.5e6  # ~ the timed region
sel = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"kernels {len(sel)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms  ({100*(span-busy)/span:.1f} %)")
per = collections.defaultdict(lambda: [0, 0])
gapafter = collections.defaultdict(lambda: [0, 0])
for a, b in zip(sel, sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    k = a["Kernel_Name"][:60]
    gapafter[k][0] += max(g, 0); gapafter[k][1] += 1
for r in sel:
    k = r["Kernel_Name"][:60]
    per[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[k][1] += 1
print("kernel, launches, total ms, avg us, idle-after total ms, avg idle-after us")
for k, (tt, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    ga = gapafter[k]
    print(f"{k:60s} {n:6d} {tt/1e6:8.2f} {tt/n/1e3:8.1f} {ga[0]/1e6:8.2f} {ga[0]/max(ga[1],1)/1e3:8.1f}")
PY
rm -rf $O/prof
tail -30 $O/r06_p_c5_gaps.txt
