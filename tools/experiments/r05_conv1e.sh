# headline step with conv1 forward's two store forms (same box, alternating)
O=gpurun_out/r05y; mkdir -p $O
L=$O/r05_y_bench_conv1_wide_ab.log
echo "## bench.py --steps 20 --warmup 3, SF_CONV1_WIDE=0 (dword stores) vs 1 (whole-line stores through LDS); LDS pitch 84, quadrow byte-offset epilogue in both" > $L
for i in 1 2; do for v in 0 1; do
  SF_CONV1_WIDE=$v python bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err
  python - $O/b_${v}_$i.json $v >> $L <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"SF_CONV1_WIDE={sys.argv[2]}", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("clock_ghz"))
for k in d["network_kernels"]["top"][:16]:
    if "conv1" in k["kernel"] or "quadrow" in k["kernel"]: print("    ", k["name"], k["kernel"], k["ms_total"], k.get("tflops"), k.get("gbps"))
P
done; done
cat $L
