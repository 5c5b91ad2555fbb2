O=gpurun_out/r6_second; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py -q -x > $O/pytest_engine.txt 2>&1; tail -3 $O/pytest_engine.txt
bash tools/runs/r6_ab.sh r6_second/ab 2 "r5:MPSE_F0_SPLIT=1 MPSE_CHOLQR_TAU=0 MPSE_CHOLQR_THETA=0" "qr:MPSE_F0_SPLIT=1" "split2:MPSE_F0_SPLIT=2" "split4:MPSE_F0_SPLIT=4" 2>&1 | tee $O/ab.txt
python -m pytest tests/test_headline_gpu.py -q -x -k "oracle or five" > $O/pytest_headline.txt 2>&1; tail -5 $O/pytest_headline.txt
