set -x
mkdir -p gpurun_out
timeout -k 10 400 python tools/run_sharded.py --bodies 1048576 --steps 1000 --energy-every 250 --body-order morton --reorder-every 50 --softening 1e-3 > gpurun_out/r04_longrun_n1048576.txt 2>&1
tail -8 gpurun_out/r04_longrun_n1048576.txt
timeout -k 10 500 python tools/run_sharded.py --bodies 4194304 --steps 100 --energy-every 50 --body-order morton --reorder-every 50 --softening 1e-3 > gpurun_out/r04_longrun_n4194304.txt 2>&1
tail -6 gpurun_out/r04_longrun_n4194304.txt
