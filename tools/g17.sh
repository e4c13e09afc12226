set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/graph_steps_ab.py 20225,8192,28749,4096 > gpurun_out/r04_graph_steps.txt 2>&1
cat gpurun_out/r04_graph_steps.txt
