set -u
O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
L=$O/r05_h_tap_perm_ab.log
export KBENCH_LAYERS=conv2 KBENCH_NS=4096,32768
echo "## conv2 forward, plain block order" > $L; python tools/kbench.py fwd >> $L 2>&1
echo "## conv2 forward, SF_TAP_PERM=1 (taps visited in groups of the four that share input elements)" >> $L; SF_TAP_PERM=1 python tools/kbench.py fwd >> $L 2>&1
echo "## repeat plain" >> $L; python tools/kbench.py fwd >> $L 2>&1
echo "## repeat SF_TAP_PERM=1" >> $L; SF_TAP_PERM=1 python tools/kbench.py fwd >> $L 2>&1
Q="--steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_events"
for V in 0 1; do
  SF_TAP_PERM=$V rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$V -o f -- python bench.py $Q > /dev/null 2> $O/f$V.err
  SF_TAP_PERM=$V rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w$V -o w -- python bench.py $Q > /dev/null 2> $O/w$V.err
  F=$(find $O/f$V -name "*counter_collection.csv" | head -1); W=$(find $O/w$V -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W > $O/r05_h_traffic_tap_perm_$V.json 2> $O/pmc$V.err
done
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
cat $L
python - <<'P'
import json
for v in (0,1):
    try:
        d=json.load(open(f"gpurun_out/r05h/r05_h_traffic_tap_perm_{v}.json"))
        ks=d.get("kernels",d)
        for k,e in ks.items():
            if "k_fwd_glds<128, 64" in k: print(v,k,e)
    except Exception as ex: print(v,"ERR",ex)
P
