set -u
(time timeout 900 python -m pytest tests/test_gpu_barostat_cases.py tests/test_gpu_rbfe_composition.py -m gpu -x -q) 2>&1 | tail -15
