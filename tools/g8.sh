set -x
mkdir -p gpurun_out
timeout -k 10 300 python tools/small_n_split.py 20225,20000,28749,8192 0 > gpurun_out/r04_small_n_final.txt 2>&1
cat gpurun_out/r04_small_n_final.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_small -o small -- python3 $GRAFT_REPO_ROOT/tools/step_trace.py 20225 one_sided 40 > $GRAFT_REPO_ROOT/gpurun_out/r04_small_pmc.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_r04_small* | head
tail -3 gpurun_out/r04_small_pmc.log
timeout -k 10 900 python -m pytest tests/ -m gpu -x -q --durations=8 > gpurun_out/r04_g8_pytest.txt 2>&1
echo "pytest rc=$?"; grep -v "^$" gpurun_out/r04_g8_pytest.txt | tail -16
