set -u
(time timeout 900 python -m pytest tests/test_gpu_rbfe_composition.py tests/test_gpu_potentials_surface.py tests/test_gpu_second_binding.py -m gpu -x -q) 2>&1 | tail -15
