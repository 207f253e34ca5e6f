O=gpurun_out/r05m; mkdir -p $O
for cfg in "SF_GLDS_ZL=0 SF_DGRAD_ZL=0 SF_TAP_PERM=0" "SF_GLDS_ZL=2 SF_DGRAD_ZL=0 SF_TAP_PERM=0" "SF_GLDS_ZL=0 SF_DGRAD_ZL=1 SF_TAP_PERM=0" "SF_GLDS_ZL=0 SF_DGRAD_ZL=0 SF_TAP_PERM=1" "SF_GLDS_ZL=2 SF_DGRAD_ZL=1 SF_TAP_PERM=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python -m pytest tests/test_gpu_parity_c2_c5.py -m gpu -q -x -k "normalize_input" 2>&1 | tail -2
  python - <<'P'
import json
d=json.load(open('gpurun_out/parity_cnn84_norm.json'))
w=d["worst_error_over_tolerance"]
print({k.split('.')[-2]+'.'+k.split('.')[-1]: (round(v['m_maxmax'],5), round(v['d_maxmax'],4), round(v['d_frac_within_2e3'],3)) for k,v in w.items() if 'weight' in k})
try:
    f=json.load(open('gpurun_out/parity_cnn84_norm_deltas_fp64.json')); print("flips", f.get("relu_flips_per_step_and_layer"))
except Exception as e: print("no fp64 file", e)
P
done
