O=gpurun_out/r6_probe; mkdir -p $O; cd $GRAFT_REPO_ROOT
python tools/headline_differ_probe.py 3 > $O/differ.txt 2>&1 &
python tools/qr_adaptive_study.py $O/qr_adaptive 6 > $O/qr_adaptive.log 2>&1
wait
tail -30 $O/differ.txt; tail -3 $O/qr_adaptive.log
