# block QR: unit tests, then the isolated timing under the environment settings given as arguments (GPU box)
# usage: bash tools/runs/qr_ab.sh <outdir> "VAR=1" "VAR=0" ...
O=${1:-gpurun_out/qrab}; shift; mkdir -p $O
python -m pytest tests/test_engine_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "qr or svd" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
for setting in "$@"; do
  echo "$setting"; env $setting python tools/qr_bench.py 2>&1 | tee $O/qr_bench_$(echo $setting | tr '= ' '__').txt
done
