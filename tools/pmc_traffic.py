"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate passes: the TCC block cannot host both) of
`bench.py --steps 1 --warmup 1` into per-kernel HBM traffic keyed by bench.py's kernel labels.

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_traffic.json

Units/corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream, so reads are doubled; WRITE_SIZE is taken as is (it matched the algorithmic output
size of every kernel here to <0.1 %).  A kernel is matched to a bench label by (op, bytes written).  The output carries
the hash of the kernel sources it was measured on (`_kernel_source_sha16`)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_factory_amd.build import source_sha16  # noqa: E402

LAYERS = {  # label-stem -> (Cin, H, W, Cout, K, S, OH, OW)
    "4x84->32": (4, 84, 84, 32, 8, 4, 20, 20), "32x20->64": (32, 20, 20, 64, 4, 2, 9, 9),
    "64x9->64": (64, 9, 9, 64, 3, 1, 7, 7), "3136x1->512": (3136, 1, 1, 512, 1, 1, 1, 1),
}


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, _ = load(sys.argv[1], "FETCH_SIZE")
    write, cnt = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for stem, (Cin, H, W, Cout, K, S, OH, OW) in LAYERS.items():
        for n in (4096, 32768):
            expect = {"fwd": n * OH * OW * Cout * 4 / 1024.0, "dgrad": n * H * W * Cin * 4 / 1024.0}
            for op, wkb in expect.items():
                if op == "dgrad" and stem.startswith("4x84"):
                    continue
                cands = [(abs(write[k] - wkb) / wkb, k) for k in write if f"k_conv_{op}<" in k[0]]
                if not cands:
                    continue
                err, k = min(cands)
                if err < 0.02 and k in fetch:
                    out[f"{op}:{stem} n={n}"] = dict(fetch_kib=round(fetch[k], 1), write_kib=round(write[k], 1),
                                                     hbm_bytes=int((2 * fetch[k] + write[k]) * 1024), launches=cnt[k],
                                                     kernel=k[0][:60], grid=k[1])
    # per kernel INSTANTIATION (bench.py's roofline object): mean HBM bytes per launch over every launch of that name
    f_all, w_all, n_all = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] == "FETCH_SIZE":
            f_all[r["Kernel_Name"]] += float(r["Counter_Value"])
            n_all[r["Kernel_Name"]] += 1
    for r in csv.DictReader(open(sys.argv[2])):
        if r["Counter_Name"] == "WRITE_SIZE":
            w_all[r["Kernel_Name"]] += float(r["Counter_Value"])
    for name, nl in n_all.items():
        if not any(t in name for t in ("k_conv", "k_fwd_glds", "k_fwd_img", "k_wgrad_glds", "k_wgrad_img", "k_dgrad_",
                                       "k_lstm_seq", "k_gru_seq", "k_linear_")):
            continue
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        out[short] = dict(fetch_kib_per_launch=round(f_all[name] / nl, 1), write_kib_per_launch=round(w_all[name] / nl, 1),
                          hbm_bytes=int((2 * f_all[name] + w_all[name]) / nl * 1024), launches=nl)
    # wgrad: one launch per layer per minibatch, identified by the layer's K*N partial size pattern (largest fetch first)
    wg = sorted(((fetch.get(k, 0), k) for k in write if "k_conv_wgrad<" in k[0]), reverse=True)
    out["_wgrad_unmatched"] = [dict(kernel=k[0][:60], grid=k[1], fetch_kib=round(f, 1), write_kib=round(write[k], 1)) for f, k in wg]
    out["_kernel_source_sha16"] = source_sha16()  # bench.py drops roofline.traffic when the kernels have changed since
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
