set -u
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/s2_tests.log 2>&1; echo "tests exit $?"; tail -4 gpurun_out/s2_tests.log
timeout 600 python bench.py --mode potentials > gpurun_out/s2_pot.json 2> gpurun_out/s2_pot.err; echo "pot exit $?"
bash scripts/gpu_stats_cmd.sh s2pot 40 python bench.py --mode potentials --systems dhfr > gpurun_out/s2_pot_stats.txt 2>&1
(time timeout 900 python bench.py) > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err; echo "bench exit $?"; tail -3 gpurun_out/s2_bench.err
