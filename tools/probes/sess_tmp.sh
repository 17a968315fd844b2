cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_s2; mkdir -p $O
timeout 900 python tests/soak_generic.py > $O/soak_generic.txt 2>&1; echo "soak_generic rc=$?" >> $O/log.txt
timeout 600 python tests/stress_mixed.py > $O/stress_mixed.txt 2>&1; echo "stress rc=$?" >> $O/log.txt
timeout 600 python tests/soak_coalesce.py > $O/soak_coalesce.txt 2>&1; echo "soak_coalesce rc=$?" >> $O/log.txt
for S in "6 17" "3 20" "20 12"; do set -- $S; timeout 900 python tests/soak.py $1 $2 > $O/soak_2pow$2.txt 2>&1; echo "soak$2 rc=$?" >> $O/log.txt; done
cat $O/log.txt; for f in soak_generic stress_mixed soak_coalesce soak_2pow17 soak_2pow20 soak_2pow12; do tail -n 1 $O/$f.txt | cut -c1-220; done
