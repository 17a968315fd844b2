cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_x; mkdir -p $O
for i in 1 2 3 4 5 6; do
  SCALE_TOTAL_PROOFS=8192 SCALE_STEPS=2 bash tools/scale_run.sh 2 dry $O/scale_$i > $O/scale_run_$i.txt 2>&1
  head -3 $O/scale_run_$i.txt | cut -c1-200
done
for i in 1 2; do
  timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu > $O/pytest_scale_$i.txt 2>&1; echo "pytest_scale_$i rc=$?"
done
