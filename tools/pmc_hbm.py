"""Achieved HBM bandwidth of the non-GEMM (HBM / latency-bound) kernels from the two TCC passes of
tools/profile_round.sh:  python tools/pmc_hbm.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json

Per kernel name: bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction for reads as in tools/pmc_traffic.py),
time = the dispatch durations recorded in the same passes; GB/s = sum(bytes) / sum(time)."""
import collections
import csv
import json
import sys

SKIP = ("k_conv", "k_fwd_glds", "k_fwd_img", "k_wgrad_glds", "k_wgrad_img", "k_dgrad_", "Cijk", "k_reduce_partials", "k_splitk_finish")


def load(path, counter):
    b, t, n = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        b[name] += float(r["Counter_Value"]) * 1024.0
        t[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n[name] += 1
    return b, t, n


def main(fetch_csv, write_csv, dst):
    fb, ft, fn = load(fetch_csv, "FETCH_SIZE")
    wb, wt, wn = load(write_csv, "WRITE_SIZE")
    out = {}
    for name in fb:
        if any(name.startswith(s) for s in SKIP) or name not in wb or fn[name] != wn[name]:
            continue
        launches = fn[name]
        byts = 2.0 * fb[name] + wb[name]
        ns = 0.5 * (ft[name] + wt[name])
        out[name] = {"launches_per_step": launches / 2.0, "avg_us": round(ns / launches / 1e3, 2),
                     "mb_per_launch": round(byts / launches / 1e6, 3), "gb_per_s": round(byts / ns, 1),
                     "frac_of_8tbs": round(byts / ns / 8000.0, 3), "us_per_step": round(ns / 2.0 / 1e3, 1)}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["us_per_step"]))
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in list(out.items())[:16]:
        print(f"{k[:60]:60s} x{v['launches_per_step']:<6.0f} {v['avg_us']:8.2f} us {v['mb_per_launch']:9.3f} MB {v['gb_per_s']:8.1f} GB/s  {v['us_per_step']:7.1f} us/step")


if __name__ == "__main__":
    main(*sys.argv[1:4])
