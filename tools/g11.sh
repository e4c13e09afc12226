set -x
mkdir -p gpurun_out
timeout -k 10 500 bash tools/profile.sh r04_pair_once > gpurun_out/r04_profile_pair_once.log 2>&1
tail -3 gpurun_out/r04_profile_pair_once.log
timeout -k 10 500 bash tools/profile.sh r04_one_sided --force-mode one_sided > gpurun_out/r04_profile_one_sided.log 2>&1
tail -3 gpurun_out/r04_profile_one_sided.log
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err
tail -c 1500 gpurun_out/r04_bench_n1.json
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1
tail -2 gpurun_out/r04_smoke.txt
