set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/ -m gpu -x -q --durations=6 > gpurun_out/r04_g10_pytest.txt 2>&1
echo "pytest rc=$?"; grep -v "^$" gpurun_out/r04_g10_pytest.txt | tail -14
