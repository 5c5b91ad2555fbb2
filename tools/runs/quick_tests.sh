cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_headline_gpu.py tests/test_tdvp_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x 2>&1 | tail -8
