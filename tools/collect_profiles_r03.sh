#!/bin/bash
# copy the judged summaries of a tools/r03_profile_session.sh run from gpurun_out/<tag>/ into profiles/ (tracked)
set -e
T=$1; S=gpurun_out/$T; P=profiles
cp $S/bench.json $P/${T}_bench.json; cp $S/bench_shard17.json $P/${T}_shard17_bench.json
cp $S/prove.json $P/${T}_prove_bench.json; cp $S/recip256.json $P/${T}_recip256_bench.json
cp $(grep -l k_verify_round $(find $S/prof -name "*kernel_stats.csv") | head -1) $P/${T}_kernel_stats.csv
cp $(find $S/prof_prove -name "*kernel_stats.csv" | head -1) $P/${T}_prove_kernel_stats.csv
cp $(find $S/prof_recip -name "*kernel_stats.csv" | head -1) $P/${T}_recip256_kernel_stats.csv
cp $S/pmc/pmc_traffic.json $P/${T}_pmc_traffic.json; cp $S/pmc/pmc_traffic.json $P/pmc_traffic.json
cp $S/sq/pmc_valu.json $P/${T}_pmc_valu.json; cp $S/sq/pmc_valu.json $P/pmc_valu.json
cp $S/box.txt $P/${T}_box.txt
[ -f $S/pytest_gpu.txt ] && cp $S/pytest_gpu.txt $P/${T}_pytest_gpu.txt
for f in soak_2pow1 soak_2pow5 soak_2pow10 soak_2pow12 soak_2pow16 soak_2pow20 stress_mixed soak_generic; do [ -s $S/$f.txt ] && tail -n 6 $S/$f.txt > $P/${T}_$f.txt; done
ls -la $P | grep ${T}_
[ -f $S/latency_w22.txt ] && cp $S/latency_w22.txt $P/${T}_latency_w22.txt; true
