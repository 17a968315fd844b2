cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_t3; mkdir -p $O
for n in 512 64 4096; do timeout 600 python tools/probes/wnla_shape_probe.py $n >> $O/shapes.txt 2>> $O/err.txt; done; echo "rc=$?" >> $O/log.txt
timeout 600 python tools/probes/latency_generic.py > $O/latency_generic.txt 2>> $O/err.txt
timeout 900 python tools/probes/recip_small_latency.py > $O/latency_recip256.txt 2>> $O/err.txt
timeout 1500 python -m pytest tests/test_gpu_wnla.py tests/test_gpu_circuit.py tests/test_gpu_recip.py tests/test_gpu_ct_generic.py tests/test_gpu_ref_fixtures.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/log.txt
cat $O/log.txt $O/shapes.txt; grep -v "kernels" $O/latency_generic.txt; grep -v kernels $O/latency_recip256.txt; tail -2 $O/pytest.txt
