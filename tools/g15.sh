set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/strip_ab.py 1048576 8 3 > gpurun_out/r04_strip_ab3.txt 2>&1
cat gpurun_out/r04_strip_ab3.txt
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "summation_parts or strips" > gpurun_out/r04_g15_pytest.txt 2>&1
echo "pytest rc=$?"; grep -v "^$" gpurun_out/r04_g15_pytest.txt | tail -4
