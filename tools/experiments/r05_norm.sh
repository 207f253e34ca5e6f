set -u
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_parity_c2_c5.py -m gpu -q -x -k "loader_fused or normalize_input" > $O/pytest_norm.log 2>&1; tail -15 $O/pytest_norm.log
B="python bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_secondary"
$B --normalize_input > $O/r05_c_bench_normalize_input.json 2> $O/b1.err; tail -c 600 $O/b1.err
SF_CONV1_NORM=0 $B --normalize_input > $O/r05_c_bench_normalize_input_materialised.json 2> $O/b2.err
$B > $O/r05_c_bench_c2.json 2> $O/b3.err
python bench.py --workload c3 --steps 6 --warmup 2 --no_cpu_baseline --no_secondary > $O/r05_c_bench_c3.json 2> $O/b4.err
SF_GLDS_SMALL64=0 SF_GLDS_SPLIT64=0 python bench.py --workload c3 --steps 6 --warmup 2 --no_cpu_baseline --no_secondary > $O/r05_c_bench_c3_old_dispatch.json 2> $O/b5.err
for f in $O/r05_c_bench_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    nk=d.get("network_kernels",{}).get("top",[])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"))
    for k in nk[:14]: print("    ", k["name"], k["kernel"], k["ms_total"], k.get("tflops"), k.get("gbs"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done
