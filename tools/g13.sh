set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/strip_ab.py 1048576 8 2 > gpurun_out/r04_strip_ab2.txt 2>&1
cat gpurun_out/r04_strip_ab2.txt
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py tests/test_body_order.py -m gpu -x -q > gpurun_out/r04_g13_pytest.txt 2>&1
echo "pytest rc=$?"; grep -v "^$" gpurun_out/r04_g13_pytest.txt | tail -6
