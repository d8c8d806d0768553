#!/bin/bash
# after the generator fixes of tests/test_gpu_random_parity.py: the test file (both bindings), the seven seeds the closing campaign
# flagged, then the same 1 500 seeds again and 1 500 fresh ones.
set -u
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
A=gpurun_out/final_campaign; mkdir -p $A
( echo "# python -m pytest tests/test_gpu_random_parity.py tests/test_gpu_second_binding.py -m gpu -q"
  timeout 900 python -m pytest tests/test_gpu_random_parity.py tests/test_gpu_second_binding.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
  for s in 20085 20148 21057 21158 21186 21355 21414; do echo "# python scripts/fuzz_parity.py $s 1"; timeout 120 python scripts/fuzz_parity.py $s 1 2>&1 | grep -v amdgpu.ids; done
  echo "# python scripts/fuzz_parity.py 20000 1500"; timeout 1200 python scripts/fuzz_parity.py 20000 1500 2>&1 | grep -v amdgpu.ids
  echo "# python scripts/fuzz_parity.py 30000 1500"; timeout 1200 python scripts/fuzz_parity.py 30000 1500 2>&1 | grep -v amdgpu.ids ) > $A/parity_recheck.txt 2>&1
cat $A/parity_recheck.txt | tail -40
