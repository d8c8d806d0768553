set -u
(time timeout 1700 python scripts/soak_rbfe.py 200000) > gpurun_out/s2_soak_rbfe.txt 2>&1; echo "exit $?"; grep -v amdgpu.ids gpurun_out/s2_soak_rbfe.txt | tail -50
