set -x
mkdir -p gpurun_out
timeout -k 10 120 tests/rccl_probe/rccl_nonblocking_probe one > gpurun_out/r04_rccl_probe.txt 2>&1
timeout -k 10 120 tests/rccl_probe/rccl_nonblocking_probe absent >> gpurun_out/r04_rccl_probe.txt 2>&1
cat gpurun_out/r04_rccl_probe.txt
timeout -k 10 600 python -m pytest tests/test_multi_process_gpu.py -m gpu -x -q -s -k "real_rccl or never_creates or never_steps or equals_the_single" > gpurun_out/r04_g4_multiproc.txt 2>&1
echo "multiproc rc=$?"; tail -12 gpurun_out/r04_g4_multiproc.txt
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "equal_mass_tiles_take or no_tile_multiple or galaxy or register_blocking or eight_row_loop or stars or bench_configuration" > gpurun_out/r04_g4_parity.txt 2>&1
echo "parity rc=$?"; tail -12 gpurun_out/r04_g4_parity.txt
timeout -k 10 300 python tools/small_n_split.py 20225,20000,16384,24576 256,0 > gpurun_out/r04_small_n_split.txt 2>&1
cat gpurun_out/r04_small_n_split.txt
timeout -k 10 400 python tools/pps_modes.py 1048576 > gpurun_out/r04_pps_modes.txt 2>&1
cat gpurun_out/r04_pps_modes.txt
