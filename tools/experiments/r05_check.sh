O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_headline_sizes.py tests/test_gpu_parity_c2_c5.py -m gpu -q -x > $O/pytest_nn.log 2>&1; tail -6 $O/pytest_nn.log
python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary > $O/r05_l_bench.json 2> $O/b.err; tail -c 300 $O/b.err
SF_GLDS_ZL=0 SF_DGRAD_ZL=0 SF_TAP_PERM=0 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary > $O/r05_l_bench_r04_kernels.json 2> $O/b2.err
python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary > $O/r05_l_bench_2.json 2> $O/b.err
for f in $O/r05_l_bench.json $O/r05_l_bench_r04_kernels.json $O/r05_l_bench_2.json; do python - "$f" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("clock_ghz"))
for k in d["network_kernels"]["top"][:14]: print("    ", k["name"], k["kernel"], k["ms_total"], k.get("tflops"))
P
done
