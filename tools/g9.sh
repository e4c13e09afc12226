set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -m gpu -x -q -k "strips" > gpurun_out/r04_g9_strips.txt 2>&1
echo "strips rc=$?"; grep -v "^$" gpurun_out/r04_g9_strips.txt | tail -12
timeout -k 10 400 python tools/strip_ab.py 1048576 8 3 > gpurun_out/r04_strip_ab.txt 2>&1
cat gpurun_out/r04_strip_ab.txt
