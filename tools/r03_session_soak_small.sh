#!/bin/bash
# round-3: long soaks of the small-call path (fresh proofs from the product prover, random corruptions, exact == RLC == CPU-oracle sample)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; OUT=gpurun_out/${1:-r03_soak_small}; mkdir -p $OUT
timeout 1200 python tests/soak.py 1500 3 > $OUT/soak_2pow3.txt 2>&1; echo "soak3 rc=$?" >> $OUT/log.txt
timeout 1200 python tests/soak.py 600 7 > $OUT/soak_2pow7.txt 2>&1; echo "soak7 rc=$?" >> $OUT/log.txt
timeout 1200 python tests/soak.py 200 11 > $OUT/soak_2pow11.txt 2>&1; echo "soak11 rc=$?" >> $OUT/log.txt
timeout 1200 python tests/soak.py 60 13 > $OUT/soak_2pow13.txt 2>&1; echo "soak13 rc=$?" >> $OUT/log.txt
cat $OUT/log.txt; tail -n 1 $OUT/soak_2pow3.txt $OUT/soak_2pow7.txt $OUT/soak_2pow11.txt $OUT/soak_2pow13.txt
