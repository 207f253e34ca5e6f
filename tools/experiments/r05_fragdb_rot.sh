# 64 x 64 wave tiles of the zero-VALU kernels: double-buffered fragment reads (SF_FRAG_DB=1 / 2), the conv2 data gradient
# with co-resident work-groups half a step out of phase (SF_QUADROW_ROT=1, key = hardware wave slot), and the "no output
# stores, MFMAs kept" ablation the r05_aa log lacks (its bit 32 let hipcc drop the MFMAs).  Same box, alternating;
# KBENCH_HASH=1: seeded inputs, a digest of every result (all builds but the ablations must print the same digests).
#   bash tools/experiments/r05_fragdb_rot.sh
O=gpurun_out/r05af; mkdir -p $O
L=$O/r05_af_fragdb_rot.log
V=$PWD/build/variants
export KBENCH_NS=32768,32768 KBENCH_LAYERS=conv2,fc KBENCH_HASH=1
echo "## tree = shipped; db1 / db2 = -DSF_FRAG_DB; rot = -DSF_QUADROW_ROT=1; nostore = -DSF_GLDS_ABLATE=32 (wrong results by design); nostore_nodma = 48" > $L
for r in 1 2; do for v in tree db1 db2 rot rotdb2 nostore nostore_nodma; do
  [ $r == 2 ] && [ $v == nostore_nodma ] && continue
  echo "lib=$v" >> $L; lib=$V/libsf_hip_$v.so; [ $v == tree ] && lib=$PWD/sample_factory_amd/libsf_hip.so
  SF_HIP_LIB=$lib timeout 300 python tools/kbench.py fwd dgrad dgrad_noact 2>&1 | grep "n= *32768" >> $L
done; done
echo "lib=tree" >> $L; timeout 300 python tools/kbench.py fwd dgrad dgrad_noact 2>&1 | grep "n= *32768" >> $L
cat $L
# headline step under the best candidate (sum of the n = 32768 launch times, digests equal to the shipped library's),
# alternating with the shipped library: 10 steps, no secondaries
BEST=$(python - $L <<'PY'
import re, sys
cur, t, dig = None, {}, {}
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("lib="): cur = l[4:]; continue
    if not l.startswith("n=") or cur is None: continue
    us = [float(x) for x in re.findall(r"([0-9.]+)us", l)]
    t.setdefault(cur, []).append(sum(us)); dig.setdefault(cur, set()).add(tuple(re.findall(r"#(\w+)", l)))
ok = [v for v in ("db1", "db2", "rot", "rotdb2") if v in t and dig[v] == dig["tree"]]
best = min(ok, key=lambda v: min(t[v])) if ok else "tree"
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
