# Tile-order / store-policy sweep of the large-tile GEMM kernels (profiles/r02_gemm_order_sweep_*.txt).  Run on the MI355X box from the
# repo root.  The two store-policy libraries are experiment builds of the GEMM file only; make them first (in the build container):
#   make -C ucod_dpl_amd/csrc variant NAME=sc1 DEFS=-DUCOD_ST_AUX=16 && make -C ucod_dpl_amd/csrc variant NAME=nt DEFS=-DUCOD_ST_AUX=2
set -x
python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -5
python tools/gemm_order_sweep.py > gpurun_out/sweep_default.txt 2>&1
UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=$PWD/ucod_dpl_amd/_native/libucod_dpl_sc1.so python tools/gemm_order_sweep.py > gpurun_out/sweep_sc1.txt 2>&1
UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=$PWD/ucod_dpl_amd/_native/libucod_dpl_nt.so python tools/gemm_order_sweep.py > gpurun_out/sweep_nt.txt 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_sweep_fetch -- python3 $R/tools/gemm_order_sweep.py --manifest $R/gpurun_out/sweep_manifest.json > /dev/null 2> $R/gpurun_out/pmc_sweep_fetch.err
python3 $R/tools/gemm_order_pmc.py $R/gpurun_out/pmc_sweep_fetch $R/gpurun_out/sweep_manifest.json FETCH_SIZE > $R/gpurun_out/sweep_fetch_default.txt 2>&1
UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=$R/ucod_dpl_amd/_native/libucod_dpl_sc1.so rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_sweep_fetch_sc1 -- python3 $R/tools/gemm_order_sweep.py --manifest $R/gpurun_out/sweep_manifest.json > /dev/null 2> $R/gpurun_out/pmc_sweep_fetch_sc1.err
python3 $R/tools/gemm_order_pmc.py $R/gpurun_out/pmc_sweep_fetch_sc1 $R/gpurun_out/sweep_manifest.json FETCH_SIZE > $R/gpurun_out/sweep_fetch_sc1.txt 2>&1
find $R/gpurun_out/pmc_sweep_fetch $R/gpurun_out/pmc_sweep_fetch_sc1 -name '*.csv' -size +2M -delete
tail -3 $R/gpurun_out/sweep_fetch_default.txt
