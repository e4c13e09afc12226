set -x
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r04_pair_once
timeout -k 10 500 bash tools/profile.sh r04_pair_once > gpurun_out/r04_profile_pair_once.log 2>&1
tail -2 gpurun_out/r04_profile_pair_once.log
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/strip_ab.py 1048576 8 2 > gpurun_out/r04_strip_ab.txt 2>&1
cat gpurun_out/r04_strip_ab.txt
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err
python -c "
import json
d=json.load(open('gpurun_out/r04_bench_n1.json'))
print({k:d[k] for k in ('value','ms_per_step','steps')}, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['reference_size']['ms_per_step'])"
