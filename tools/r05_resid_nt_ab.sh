# NOTE (round 6): the environment knobs this tool sets are honoured by the -DUCOD_LAB_KNOBS builds only: `make -C ucod_dpl_amd/csrc knobs`, then run with
#   UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=ucod_dpl_amd/_native/libucod_dpl_knobs.so UCOD_DPL_EXPERIMENT_LIB_F16=ucod_dpl_amd/_native/libucod_dpl_f16_knobs.so
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in 0 1; do
    UCOD_RESID16_NT=$v timeout 300 python bench.py --no-cpu-baseline --lora-steps 0 --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('nt=$v', d['value'], d['ms_per_step'], 'projfc2', k['gemm_bf16_proj_fc2_scale_resid']['avg_us'], 'qkv', k['gemm_bf16_qkv_bias']['avg_us'], 'fc1', k['gemm_bf16_fc1_gelu']['avg_us'], 'attn', k['attention_fwd']['avg_us'], 'logit', d['logit_max_abs'])"
  done
done
