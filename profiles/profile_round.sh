#!/bin/bash
# Run on the GPU box (gpurun -- 'bash profiles/profile_round.sh r01'): the bench line, the
# rocprofv3 kernel trace of the same command and the two PMC passes behind roofline.traffic.
# Outputs under gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards.
set -e
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $root/bench.py --steps 3 --warmup 1 > $out/bench.json 2> $out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
python3 $root/profiles/pmc_summary.py $out/pmc_fetch_write.json FETCH_SIZE=$out/pmc_fetch WRITE_SIZE=$out/pmc_write
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
ls $out
