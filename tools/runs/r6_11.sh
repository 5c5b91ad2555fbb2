O=gpurun_out/r6_11; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/runs/r6_ab.sh r6_11/ab 2 "cap512:MPSE_RED_MAX_BLOCKS=512" "cap1024:MPSE_RED_MAX_BLOCKS=1024" "cap2048:MPSE_RED_MAX_BLOCKS=2048" 2>&1 | tee $O/ab.txt
python -m pytest tests/test_engine_gpu.py tests/test_tdvp_gpu.py tests/test_dmrg_gpu.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
