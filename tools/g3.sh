set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/ -m gpu -x -q --durations=15 > gpurun_out/r04_g3_pytest.txt 2>&1
echo "pytest rc=$?"
tail -40 gpurun_out/r04_g3_pytest.txt
