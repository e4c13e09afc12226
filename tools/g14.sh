set -x
mkdir -p gpurun_out
timeout -k 10 1100 python tests/fuzz_gpu.py 120 404 > gpurun_out/r04_fuzz.txt 2>&1
echo "fuzz rc=$?"; tail -6 gpurun_out/r04_fuzz.txt
