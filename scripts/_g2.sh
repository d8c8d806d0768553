set -u
R=$GRAFT_REPO_ROOT
bash scripts/gpu_stats_cmd.sh s2pp64 6 python $R/scripts/pp_launch_probe.py f64 | cut -c1-200
bash scripts/gpu_stats_cmd.sh s2pp32 6 python $R/scripts/pp_launch_probe.py f32 | cut -c1-200
(time timeout 900 python -m pytest tests -m gpu -x -q) 2>&1 | tail -6
