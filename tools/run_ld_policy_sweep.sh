# Operand-load cache-policy sweep of the large-tile GEMM (profiles/r02_gemm_load_policy_*.txt).  Experiment builds first:
#   make -C ucod_dpl_amd/csrc variant NAME=lda DEFS=-DUCOD_LD_AUX_A=2; ... NAME=ldb DEFS=-DUCOD_LD_AUX_B=2; ... NAME=ldab DEFS="-DUCOD_LD_AUX_A=2 -DUCOD_LD_AUX_B=2"
R=$PWD
for v in default lda ldb ldab; do
  L=$R/ucod_dpl_amd/_native/libucod_dpl.so; [ $v != default ] && L=$R/ucod_dpl_amd/_native/libucod_dpl_$v.so
  UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=$L python tools/gemm_order_sweep.py > gpurun_out/ldsweep_$v.txt 2>&1
done
cd /tmp; export TMPDIR=/tmp
for v in default lda ldab; do
  L=$R/ucod_dpl_amd/_native/libucod_dpl.so; [ $v != default ] && L=$R/ucod_dpl_amd/_native/libucod_dpl_$v.so
  UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_ld_$v -- python3 $R/tools/gemm_order_sweep.py --manifest $R/gpurun_out/ld_manifest.json > /dev/null 2> $R/gpurun_out/pmc_ld_$v.err
  python3 $R/tools/gemm_order_pmc.py $R/gpurun_out/pmc_ld_$v $R/gpurun_out/ld_manifest.json FETCH_SIZE > $R/gpurun_out/ldsweep_fetch_$v.txt 2>&1
  find $R/gpurun_out/pmc_ld_$v -name '*.csv' -size +2M -delete
done
tail -4 $R/gpurun_out/ldsweep_fetch_lda.txt
