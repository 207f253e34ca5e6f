O=gpurun_out/r05p; mkdir -p $O
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary > $O/r05_p_bench_$i.json 2> $O/b$i.err; done
SF_GLDS_ZL=0 SF_DGRAD_ZL=0 SF_WGRAD_ZL=0 python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary > $O/r05_p_bench_r04_kernels.json 2> $O/b0.err
for f in $O/r05_p_bench_1.json $O/r05_p_bench_2.json $O/r05_p_bench_r04_kernels.json; do python - "$f" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("clock_ghz"))
for k in d["network_kernels"]["top"][:15]: print("    ", k["name"], k["kernel"], k["ms_total"], k.get("tflops"))
P
done
