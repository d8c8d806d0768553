set -u
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pregather_with_atom_subset") 2>&1 | tail -30
