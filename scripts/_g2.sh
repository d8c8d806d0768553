timeout 900 python bench.py --mode potentials > gpurun_out/s3_pot.json 2> gpurun_out/s3_pot.err; echo "exit $?"; uptime
