set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/small_n_split.py 20225,20000,14791,28749 256,0 > gpurun_out/r04_small_n_split.txt 2>&1
cat gpurun_out/r04_small_n_split.txt
timeout -k 10 700 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -s -k "equal_mass_tiles_take or no_tile_multiple or galaxy or register_blocking or stars or bench_configuration or k17 or per_particle" > gpurun_out/r04_g5_parity.txt 2>&1
echo "parity rc=$?"; grep -v "^$" gpurun_out/r04_g5_parity.txt | tail -25
