#!/bin/bash
# Everything under profiles/<round>/ for the current build, on the GPU box from the repo root:
#   bash profiles/run_all.sh gpurun_out/prof <suffix>
# kernel stats + timeline + launch counts of the headline workload, PMC passes + summary, the small workloads, the long-video run.
OUT=${1:-gpurun_out/prof}; SFX=${2:-b}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks
cd $ROOT
# SMIN_SYNC_WEIGHTS=1: the weight-gradient stream folded into the main one, so that a kernel's duration is the kernel's
SMIN_SYNC_WEIGHTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-modes > $OUT/bench_under_rocprof_$SFX.log 2>&1
f=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1)
cp $f $OUT/bench_kernel_stats_$SFX.csv
python profiles/summarize.py $f 23 70 > $OUT/bench_kernel_stats_$SFX.txt
python tools/timeline.py $(find /tmp/prof_ks -name "*kernel_trace.csv" | head -1) > $OUT/bench_timeline_serial_$SFX.txt 2>&1
bash tools/launches.sh anet_$SFX && cp gpurun_out/tl_anet_$SFX.txt $OUT/bench_timeline_$SFX.txt && cp gpurun_out/launches_anet_$SFX.txt $OUT/bench_launches_$SFX.txt
bash tools/launches.sh ch_$SFX --workload charadessta && cp gpurun_out/tl_ch_$SFX.txt $OUT/charadessta_timeline_$SFX.txt && cp gpurun_out/launches_ch_$SFX.txt $OUT/charadessta_launches_$SFX.txt
bash tools/launches.sh ta_$SFX --workload tacos && cp gpurun_out/tl_ta_$SFX.txt $OUT/tacos_timeline_$SFX.txt && cp gpurun_out/launches_ta_$SFX.txt $OUT/tacos_launches_$SFX.txt
SMIN_SYNC_WEIGHTS=1 bash profiles/run_pmc.sh $OUT/pmc_$SFX > $OUT/pmc_$SFX.log 2>&1
python profiles/pmc_summary.py $OUT/pmc_$SFX > $OUT/pmc_top_$SFX.txt 2>&1
cp $OUT/pmc_$SFX/pmc_summary.json $OUT/pmc_summary_$SFX.json
python profiles/make_pmc_moment.py $OUT/pmc_summary_$SFX.json > $OUT/pmc_moment_$SFX.json
rm -rf $OUT/pmc_$SFX
# long-video stress (BASELINE configs[4] at one GPU's share): kernel stats
rm -rf /tmp/prof_lv
cd /tmp
SMIN_SYNC_WEIGHTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lv -- python3 $ROOT/bench.py --workload longvideo --steps 4 --warmup 2 --no-cpu-baseline --no-other-modes > $ROOT/$OUT/longvideo_under_rocprof_$SFX.log 2>&1
cd $ROOT
python profiles/summarize.py $(find /tmp/prof_lv -name "*kernel_stats.csv" | head -1) 6 40 > $OUT/longvideo_kernel_stats_$SFX.txt
python bench.py --workload longvideo --steps 5 --warmup 2 --no-cpu-baseline > $OUT/longvideo_bench_$SFX.json 2>/dev/null
