# the driver's entry points on a fresh box: smoke(), the default bench line, the 2-rank launch as the driver invokes it (gloo: one GPU)
O=gpurun_out/r06u; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
(time python bench.py > $O/bench_default.json 2> $O/bench_default.err) 2>&1 | grep real | tee -a $O/smoke.log
python -c "
import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['steps'],d['warmup'],d['roofline']['frac'],d['roofline']['traffic'],len(d['secondary']),d['cpu_baseline']['value'])" | tee -a $O/smoke.log
SF_DP_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 2 --envs 512 2> $O/g2.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('gpus2', d['n_gpus'], d['value'], d['ms_per_step'], d.get('rank_affinity'), d.get('collectives',{}).get('exposed_ms_per_step'), d.get('collectives',{}).get('small_buckets'))" | tee -a $O/smoke.log
SF_DP_ONESHOT=1 SF_DP_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 2 --envs 512 2> $O/g2os.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('gpus2 oneshot', d['n_gpus'], d['value'], d['ms_per_step'], d.get('collectives',{}).get('exposed_ms_per_step'), d.get('collectives',{}).get('small_buckets'))" | tee -a $O/smoke.log
tail -3 $O/g2os.err
