#!/bin/bash
# copy the judged summaries of a tools/session.sh run from gpurun_out/<tag>/ into profiles/ (tracked); whatever the session made
# usage: tools/collect_profiles.sh <tag> [also-as-current]   ("also-as-current": the PMC summaries become profiles/pmc_{traffic,valu}.json,
# the files bench.py reads -- do that only for a session that ran the build being shipped)
set -u
T=$1; S=gpurun_out/$T; P=profiles/${T%%_*}; mkdir -p $P      # profiles/r05/ for a session tag r05_x
TOP=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2"; }
cpif $S/bench.json $P/${T}_bench.json; cpif $S/bench_shard17.json $P/${T}_shard17_bench.json
cpif $S/prove.json $P/${T}_prove_bench.json; cpif $S/recip256.json $P/${T}_recip256_bench.json
for d in prof prof_prove prof_recip; do
  f=$(find $S/$d -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $P/${T}_${d/prof/kernel_stats}.csv
done
[ -f $P/${T}_kernel_stats_prove.csv ] && mv $P/${T}_kernel_stats_prove.csv $P/${T}_prove_kernel_stats.csv
[ -f $P/${T}_kernel_stats_recip.csv ] && mv $P/${T}_kernel_stats_recip.csv $P/${T}_recip256_kernel_stats.csv
cpif $S/pmc/pmc_traffic.json $P/${T}_pmc_traffic.json; cpif $S/sq/pmc_valu.json $P/${T}_pmc_valu.json
if [ "${2:-}" = "also-as-current" ]; then cpif $S/pmc/pmc_traffic.json $TOP/pmc_traffic.json; cpif $S/sq/pmc_valu.json $TOP/pmc_valu.json; fi
cpif $S/box.txt $P/${T}_box.txt; cpif $S/log.txt $P/${T}_log.txt
cpif $S/pytest_gpu.txt $P/${T}_pytest_gpu.txt; cpif $S/smoke.txt $P/${T}_smoke.txt
cpif $S/concurrent_callers.json $P/${T}_concurrent_callers.json; cpif $S/concurrent_callers_prove.json $P/${T}_concurrent_callers_prove.json; cpif $S/concurrent_callers_prove.txt $P/${T}_concurrent_callers_prove.txt
for f in soak_2pow1 soak_2pow5 soak_2pow10 soak_2pow12 soak_2pow16 soak_2pow20 stress_mixed soak_generic; do [ -s $S/$f.txt ] && tail -n 6 $S/$f.txt > $P/${T}_$f.txt; done
cpif $S/latency_w22.txt $P/${T}_latency_w22.txt
for f in $S/cmd_*.txt; do [ -s "$f" ] && cp "$f" $P/${T}_$(basename $f); done
ls $P | grep "^${T}_"
