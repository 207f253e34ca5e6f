"""Sum rocprofv3 --pmc counters per kernel name.  python tools/pmc_dump.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    n[name].add(r["Dispatch_Id"])
for name, c in agg.items():
    print(name, "launches", len(n[name]))
    for k, v in sorted(c.items()):
        print(f"   {k:28s} {v / len(n[name]):16.0f} per launch")
