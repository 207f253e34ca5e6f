"""Per-kernel sums of the SQ counters of tools/stall_counters.sh (two rocprofv3 counter_collection.csv files) and the
ratios that say where a wave's cycles go: python tools/stall_counters.py <pass1.csv> <pass2.csv>"""
import csv
import re
import sys
from collections import defaultdict

KEEP = ("k_fwd_glds", "k_fwd_img", "k_dgrad", "k_wgrad", "k_conv1", "k_lstm_seq", "k_gru_seq")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name).strip()


def load(path):
    out = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    if not path:
        return out, launches
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if not any(t in k for t in KEEP):
                continue
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    return out, launches


a, la = load(sys.argv[1] if len(sys.argv) > 1 else "")
b, _ = load(sys.argv[2] if len(sys.argv) > 2 else "")
for k in sorted(a, key=lambda k: -a[k].get("SQ_WAVE_CYCLES", 0)):
    c, d = a[k], b.get(k, {})
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
    wc2 = d.get("SQ_WAVE_CYCLES", 0) or 1.0
    f = lambda v, w: f"{v / w:5.2f}"
    print(f"{k:46s} launches {len(la[k]):3d}  of wave cycles: wait_any {f(c.get('SQ_WAIT_ANY', 0), wc)}  wait_inst_any "
          f"{f(c.get('SQ_WAIT_INST_ANY', 0), wc)}  wait_inst_lds {f(c.get('SQ_WAIT_INST_LDS', 0), wc)}  active_inst_any "
          f"{f(c.get('SQ_ACTIVE_INST_ANY', 0), wc)} | active valu {f(d.get('SQ_ACTIVE_INST_VALU', 0), wc2)}  lds "
          f"{f(d.get('SQ_ACTIVE_INST_LDS', 0), wc2)}  vmem {f(d.get('SQ_ACTIVE_INST_VMEM', 0), wc2)}  scalar "
          f"{f(d.get('SQ_ACTIVE_INST_SCA', 0), wc2)}  misc {f(d.get('SQ_ACTIVE_INST_MISC', 0), wc2)}  "
          f"lds_bank_conflict/lds_active {d.get('SQ_LDS_BANK_CONFLICT', 0) / (d.get('SQ_ACTIVE_INST_LDS', 0) or 1.0):5.2f}")
