#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
A=gpurun_out/final_campaign; mkdir -p $A
( echo "# python -m pytest tests/test_gpu_random_parity.py tests/test_gpu_second_binding.py -m gpu -q"
  timeout 900 python -m pytest tests/test_gpu_random_parity.py tests/test_gpu_second_binding.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
  for s in 20085 20295 21057 21186 21414 31026 31377 20148 21158 21355; do echo "# python scripts/fuzz_parity.py $s 1"; timeout 120 python scripts/fuzz_parity.py $s 1 2>&1 | grep -v amdgpu.ids; done
  echo "# python scripts/fuzz_parity.py 40000 600"; timeout 900 python scripts/fuzz_parity.py 40000 600 2>&1 | grep -v amdgpu.ids ) > $A/parity_recheck3.txt 2>&1
cat $A/parity_recheck3.txt | tail -30
