#!/bin/bash
# Run on the GPU box (gpurun -- 'bash profiles/profile_round.sh r03 20 5 [c3|c2|c4|c5]'): the bench
# line of the DRIVER's command (--steps 20 --warmup 5, config c3 by default), the rocprofv3 kernel
# trace of the same command, and separate PMC passes of the same command: FETCH_SIZE, WRITE_SIZE (roofline.traffic)
# and the FP64 MFMA counters. Outputs under gpurun_out/prof_<tag>/; the summaries are copied into
# profiles/ by hand afterwards (gpurun_out/ is scratch).
tag=${1:-r02}
steps=${2:-20}
warm=${3:-5}
cfg=${4:-c3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_${tag}_$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --config $cfg --steps $steps --warmup $warm"
timeout -k 10 600 $cmd > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $cmd --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
rocprofv3 -L > $out/counters_available.txt 2>&1
for pass in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -o pmc -- $cmd --no-cpu-baseline > $out/pmc_$name.json 2> $out/pmc_$name.err || echo "pass $name failed" >> $out/failed.txt
done
python3 $root/profiles/pmc_summary.py $out/pmc_${cfg}_${steps}_${warm}.json FETCH_SIZE=$out/pmc_FETCH_SIZE WRITE_SIZE=$out/pmc_WRITE_SIZE \
  SQ_INSTS_VALU_MFMA_MOPS_F64=$out/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES=$out/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64 \
  SQ_BUSY_CYCLES=$out/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES=$out/pmc_SQ_INSTS_VALU_MFMA_MOPS_F64
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
(cd $root/profiles && python3 mfma_util.py $out $out/mfma_util.json $cfg $steps $warm)
# compact per-launch timeline (start us, duration us, stream, short kernel name) of the traced run
python3 - $out <<'PYEOF'
import csv, glob, sys
out = sys.argv[1]
files = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)
if files:
    rows = list(csv.DictReader(open(files[0])))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(out + "/timeline_compact.csv", "w") as fh:
        fh.write("start_us,dur_us,stream,kernel\n")
        for r in rows:
            name = r["Kernel_Name"].replace("magma::", "").split("(")[0][:48]
            fh.write("%.1f,%.1f,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Stream_Id", ""), name))
PYEOF
# keep the merge small: the raw traces stay on the box
find $out -name "*.csv" -size +1M -delete
find $out -name "*.db" -delete
ls $out
