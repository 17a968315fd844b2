cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_q3; mkdir -p $O
for rep in 1 2; do for n in 32768 65536; do for B in 0 1; do
  BPPP_GENERIC_FB_BLOCKS=$B timeout 600 python bench.py --workload recip256 --total-proofs $n --no-cpu-baseline --steps 10 > $O/r.json 2> $O/r.err
  python - <<P >> $O/ab.txt
import json
d=json.loads(open("$O/r.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
print("n=$n blocks=$B", round(d["value"]), round(d["ms_per_step"],3), d.get("accept_bits_ok"), "c0_fixed", round(k["k_recip_c0_fixed"],3), "msm", round(k["k_wnla_msm"],3))
P
done; done; done
cat $O/ab.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-session-rates --steps 5 > $O/u64.json 2> $O/u64.err; echo "u64 rc=$?" >> $O/log.txt
python tools/show_bench.py $O/u64.json | head -4
timeout 1200 python -m pytest tests/test_gpu_recip.py tests/test_gpu_wnla.py tests/test_gpu_circuit.py tests/test_gpu_scale.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/log.txt
cat $O/log.txt; tail -3 $O/pytest.txt
