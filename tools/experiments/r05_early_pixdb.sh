# third pass: tree = conv2 data gradient with double-buffered fragments (SF_QUADROW_FRAG_DB=2, the new default), q0 = without;
# early = -DSF_MID_EARLY=1 (64 x 32 wave tiles: next group's fragments + DMA behind the group's first k-step: conv2 forward, the
# rollout-size fc forward); pixdb = -DSF_PIX_FRAG_DB=1 (conv3 data gradient with double-buffered fragments); all = both
#   bash tools/experiments/r05_early_pixdb.sh
O=gpurun_out/r05ah; mkdir -p $O
L=$O/r05_ah_early_pixdb.log
V=$PWD/build/variants
export KBENCH_NS=4096,32768 KBENCH_LAYERS=conv2,conv3,fc KBENCH_HASH=1
echo "## tree = shipped; q0 = -DSF_QUADROW_FRAG_DB=0; early = -DSF_MID_EARLY=1; pixdb = -DSF_PIX_FRAG_DB=1; all = early + pixdb" > $L
for r in 1 2; do for v in tree q0 early pixdb all; do
  echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib timeout 300 python tools/kbench.py fwd dgrad_noact 2>&1 | grep "^n=" >> $L
done; done
echo "lib=tree" >> $L; timeout 300 python tools/kbench.py fwd dgrad_noact 2>&1 | grep "^n=" >> $L
cat $L
BEST=$(python - $L <<'PY'
import re, sys
cur, t, dig = None, {}, {}
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("lib="):
        cur = l[4:]; t.setdefault(cur, []).append(0.0); continue
    if not l.startswith("n=") or cur is None: continue
    w = 33 if l.startswith("n=  4096") else 4   # launches per step
    us = re.findall(r"(fwd_t|dgrad\(no act read\)) +([0-9.]+)us", l)
    t[cur][-1] += w * sum(float(x[1]) for x in us)
    dig.setdefault(cur, set()).add(tuple(re.findall(r"#(\w+)", l)))
ok = [v for v in ("early", "pixdb", "all") if v in t and dig[v] == dig["tree"]]
for v in t: print(v, [round(x) for x in t[v]], "digests equal" if dig[v] == dig["tree"] else "DIGESTS DIFFER", file=sys.stderr)
best = min(ok, key=lambda v: sum(t[v]) / len(t[v])) if ok else "tree"
print(best)
PY
)
echo "best candidate with identical digests: $BEST" | tee -a $L
Q="--steps 10 --warmup 3 --no_cpu_baseline --no_secondary"
for v in tree $BEST tree $BEST; do
  lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  echo "bench lib=$v" | tee -a $L
  SF_HIP_LIB=$lib timeout 300 python bench.py $Q 2>$O/bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))" | tee -a $L
done
