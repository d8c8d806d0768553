set -u
R=$GRAFT_REPO_ROOT
for a in "dhfr f64 same" "dhfr f32 same" "config5 f64 same" "config5 f64 windows" "config5 f32 windows"; do python scripts/further_sets_probe.py $a 2>&1 | grep -v amdgpu.ids; done
bash scripts/gpu_stats_cmd.sh s2fs1 12 python $R/scripts/further_sets_probe.py dhfr f64 same
