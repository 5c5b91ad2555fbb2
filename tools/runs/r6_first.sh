# round 6, first pass: changed tests + start-of-round headline on this round's box
O=gpurun_out/r6_first; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py -q -x -k "block_qr" > $O/pytest_qr.txt 2>&1; tail -5 $O/pytest_qr.txt
python -m pytest tests/test_headline_gpu.py -q -x -k "oracle or five or bond_dims" > $O/pytest_headline.txt 2>&1; tail -5 $O/pytest_headline.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; tail -c 600 $O/bench20.json
