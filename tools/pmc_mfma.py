"""Per-kernel matrix-pipe utilisation from one rocprofv3 SQ counter pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY \
              SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc -o p -- \
              python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events
    python tools/pmc_mfma.py gpurun_out/pmc/p_counter_collection.csv profiles/rNN_mfma_util.json

Units (checked against the instruction counts of the same pass):
  * SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA is exactly 64 for the v_mfma_f32_32x32x2_f32 kernels and 32 for the
    v_mfma_f32_16x16x4_f32 ones, i.e. the counter is matrix-pipe cycles summed over all SIMDs;
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs (GUI / wall-ns = 18.4 = 8 x 2.3 GHz), so the gfx94x MfmaUtil formula
    busy / (GUI * CUs * 4) under-reports by 8x on this part; the per-XCD clock count is GUI / 8.
  mfma_util = busy / ((GUI / 8) * 256 CUs * 4 SIMDs)
"""
import collections
import csv
import json
import sys

XCDS, CUS, SIMDS = 8, 256, 4
KEEP = ("k_conv", "k_fwd_glds", "k_fwd_img", "k_wgrad_glds", "k_wgrad_img", "k_dgrad_", "k_lstm_seq", "k_gru_seq")


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    wall_ns = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            launches[name] += 1
            wall_ns[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for name, c in agg.items():
        if not any(t in name for t in KEEP) or c["GRBM_GUI_ACTIVE"] <= 0 or c["SQ_INSTS_MFMA"] <= 0:
            continue
        clk = c["GRBM_GUI_ACTIVE"] / XCDS
        out[name] = {
            "launches": launches[name],
            "avg_us_profiled": round(wall_ns[name] / launches[name] / 1e3, 1),
            "clock_ghz": round(clk / wall_ns[name], 3),
            "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (clk * CUS * SIMDS), 3),
            "busy_cycles_per_mfma": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"], 1),
            "valu_per_mfma": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2),
            "lds_per_mfma": round(c["SQ_INSTS_LDS"] / c["SQ_INSTS_MFMA"], 2),
            "issue_stall_frac": round(c["SQ_WAIT_INST_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]), 3),
        }
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_us_profiled"]))
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:42s} x{v['launches']:<4d} {v['avg_us_profiled']:8.1f} us  mfma_util {v['mfma_util']:.3f}  "
              f"valu/mfma {v['valu_per_mfma']:5.2f}  lds/mfma {v['lds_per_mfma']:4.2f}  stall {v['issue_stall_frac']:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
