#!/bin/bash
# GPU box: rocprofv3 summaries and bench lines for profiles/ (tag = $1).  Output under gpurun_out/<tag>/.
# Order: kernel trace of the default bench command, the two PMC passes (condensed into profiles/<tag>_pmc_fetch_write.csv, which
# bench.py reads for roofline.traffic), then the bench lines of the three configurations and the GPU test log.
set -e
TAG=${1:-r03}
PART=${2:-all}    # all | stats | pmc | bench: a GPU call is limited to 20 minutes, the parts can be run one per call
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--no-cpu-baseline --no-variants"
if [ $PART = all ] || [ $PART = stats ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $FAST > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/stats -name "*kernel_trace.csv" | head -1) $O/kernel_trace_full.csv
python3 $R/scripts/lane_overlap.py $O/kernel_trace_full.csv $O/trace_excerpt.csv > $O/lane_overlap.txt 2>&1 || true
rm -rf $O/stats $O/kernel_trace_full.csv
echo "stats done"
fi
if [ $PART = all ] || [ $PART = pmc ]; then
# the synthetic stream is rendered by a pool of forked workers; under a counter pass the profiler's library lives in every forked
# child and its SIGTERM handler stalls when the pool ends -- render in-process there
export RDVIO_SYNTH_WORKERS=1
# (a counter pass that stalls is cut off and tried once more; the summary is written from whatever passes completed)
for C in FETCH_SIZE WRITE_SIZE; do
    D=$O/pmc_$(echo $C | cut -d_ -f1 | tr A-Z a-z)
    for TRY in 1 2; do
        rm -rf $D
        if timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/bench.py $FAST --no-kernel-loop --steps 60 --warmup 5 > $D.log 2>&1; then
            echo "$C done (try $TRY)"
            break
        fi
        echo "$C pass failed or stalled (try $TRY)"
    done
done
mkdir -p $O/pmc_fetch $O/pmc_write
python3 $R/scripts/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_all.csv > /dev/null
grep -v "^__amd\|^at::" $O/pmc_all.csv > $O/pmc_fetch_write.csv
cp $O/pmc_fetch_write.csv $R/profiles/${TAG}_pmc_fetch_write.csv
# bench.py reads profiles/<round>_pmc_fetch_write.csv (+ .meta.json): refresh it in place so that the bench lines below carry `traffic`
ROUND=${TAG%%_*}
cp $O/pmc_fetch_write.csv $R/profiles/${ROUND}_pmc_fetch_write.csv
python3 -c "import sys, json; sys.path.insert(0, '$R'); import bench; json.dump({'kernel_source_sha256': bench.kernel_source_hash(), 'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --no-cpu-baseline --no-variants --no-kernel-loop --steps 60 --warmup 5'}, open('$O/pmc_fetch_write.meta.json', 'w'))"
cp $O/pmc_fetch_write.meta.json $R/profiles/${TAG}_pmc_fetch_write.meta.json
cp $O/pmc_fetch_write.meta.json $R/profiles/${ROUND}_pmc_fetch_write.meta.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_all.csv
unset RDVIO_SYNTH_WORKERS
echo "pmc done"
fi
if [ $PART = all ] || [ $PART = bench ]; then
cd $R
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
# configs 3 and 5 WITH the CPU legs: feature_indices_identical / ate_rmse_gpu_vs_cpu_path_mm exist for every config
python3 $R/bench.py --config synthetic_720p --steps 200 --warmup 10 > $O/bench_synthetic_720p.json 2>> $O/bench.err
echo "720p done"
python3 $R/bench.py --config euroc_mh03_rd --steps 300 --warmup 10 > $O/bench_euroc_mh03_rd.json 2>> $O/bench.err
echo "mh03 done"
RDVIO_PIPELINE_PROF=1 python3 $R/scripts/pipeline_fps.py --frames 400 --modes 2 --bootstrap init > $O/pipeline_phases.log 2>&1
python3 -m pytest $R/tests -m gpu -q > $O/gpu_tests.log 2>&1 || true
fi
echo "all done"
