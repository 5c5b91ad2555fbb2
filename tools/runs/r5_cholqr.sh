# Cholesky-QR path: correctness cases, timing against the Householder path, per-kernel profile, the QR tests of the suite
T=${1:-r5_cholqr}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 300 python tools/cholqr_check.py $O/cholqr_check.md > $O/cholqr_check.out 2>&1; echo "exit $?" >> $O/cholqr_check.out)
(timeout 300 python tools/qr_bench.py > $O/qr_bench_chol.txt 2>&1); (MPSE_CHOLQR=0 timeout 300 python tools/qr_bench.py > $O/qr_bench_hh.txt 2>&1)
bash tools/qr_profile.sh $O > /dev/null 2>&1
if [ "$2" = "tests" ]; then (timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "qr or svd" > $O/pytest_qr.txt 2>&1); tail -3 $O/pytest_qr.txt; fi
cat $O/cholqr_check.out; paste $O/qr_bench_chol.txt $O/qr_bench_hh.txt; head -12 $O/qr_kernels_256x16x256.md; head -8 $O/qr_kernels_256x2x256.md
