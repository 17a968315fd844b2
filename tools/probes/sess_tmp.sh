cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_u; mkdir -p $O
for S in "12 17" "8 18" "6 20" "40 16" "100 10"; do set -- $S; timeout 1200 python tests/soak.py $1 $2 > $O/soak_2pow$2.txt 2>&1; echo "soak$2 rc=$?" >> $O/log.txt; done
timeout 900 python tests/stress_mixed.py > $O/stress_mixed.txt 2>&1; echo "stress rc=$?" >> $O/log.txt
timeout 900 python tests/soak_generic.py > $O/soak_generic.txt 2>&1; echo "soak_generic rc=$?" >> $O/log.txt
timeout 900 python tests/soak_coalesce.py > $O/soak_coalesce.txt 2>&1; echo "soak_coalesce rc=$?" >> $O/log.txt
cat $O/log.txt; for f in soak_2pow17 soak_2pow18 soak_2pow20 stress_mixed soak_generic soak_coalesce; do tail -n 2 $O/$f.txt | cut -c1-200; done
