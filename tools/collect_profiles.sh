#!/bin/bash
# copy the judged summaries of a tools/r02_profile_session.sh run from gpurun_out/<tag>/ into profiles/ (tracked)
# usage: tools/collect_profiles.sh <tag>
set -e
T=$1; S=gpurun_out/$T; P=profiles
cp $S/bench.json $P/${T}_bench.json
cp $S/prove.json $P/${T}_prove_bench.json
cp $S/recip256.json $P/${T}_recip256_bench.json
cp $(find $S/prof -name "*kernel_stats.csv" | head -1) $P/${T}_kernel_stats.csv
cp $(find $S/prof_prove -name "*kernel_stats.csv" | head -1) $P/${T}_prove_kernel_stats.csv
cp $(find $S/prof_recip -name "*kernel_stats.csv" | head -1) $P/${T}_recip256_kernel_stats.csv
cp $S/pmc/pmc_traffic.json $P/${T}_pmc_traffic.json; cp $S/pmc/pmc_traffic.json $P/pmc_traffic.json
cp $S/sq/pmc_valu.json $P/${T}_pmc_valu.json; cp $S/sq/pmc_valu.json $P/pmc_valu.json
[ -f $S/pytest_gpu.txt ] && cp $S/pytest_gpu.txt $P/${T}_pytest_gpu.txt
[ -s $S/rlc_sweep.jsonl ] && cp $S/rlc_sweep.jsonl $P/${T}_rlc_sweep.jsonl
ls -la $P | grep ${T}_
