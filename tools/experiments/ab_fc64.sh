# A/B of the rollout-size fc forward plans (SF_GLDS_FC64: 0 = 128x128 tiles x 4 K slices + k_splitk_finish, 1 = 64x64
# tiles unsplit, 2 = 128x64 tiles x 2 K slices); run from the repo root on the GPU box
mkdir -p gpurun_out/r04_d; O=gpurun_out/r04_d/fc64_ab2.log; : > $O
for i in 1 2 3; do for v in 0 1 2; do echo "SF_GLDS_FC64=$v run $i" >> $O
SF_GLDS_FC64=$v python bench.py --steps 20 --no_cpu_baseline --no_secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step']}))
for k in d['network_kernels']['top']:
    if 'n=4096' in k['kernel'] and '3136' in k['kernel']: print('  ',k)
" >> $O; done; done
cat $O
