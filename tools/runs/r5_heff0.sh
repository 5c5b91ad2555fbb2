O=gpurun_out/${1:-r5_heff0}; mkdir -p $O; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "fused or expm or heff" > $O/pytest.txt 2>&1); tail -25 $O/pytest.txt
