#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03f_pytest.txt 2>&1
tail -6 gpurun_out/r03f_pytest.txt
