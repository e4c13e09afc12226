set -x
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_multi_process_gpu.py -m gpu -x -q -s > gpurun_out/r04_g2_multiproc.txt 2>&1
echo "multiproc rc=$?"
tail -15 gpurun_out/r04_g2_multiproc.txt
timeout -k 10 600 python tools/edge_mutations.py > gpurun_out/r04_edge_mutations.txt 2>&1
echo "mutations rc=$?"
tail -60 gpurun_out/r04_edge_mutations.txt
