#!/bin/bash
# GPU box: bench lines + rocprofv3 summaries for profiles/ (tag = $1).  Output under gpurun_out/<tag>/.
set -e
TAG=${1:-r01_d}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --end-to-end-frames 0 > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --no-cpu-baseline --end-to-end-frames 0 --steps 30 --warmup 5 > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --no-cpu-baseline --end-to-end-frames 0 --steps 30 --warmup 5 > $O/pmc_write.log 2>&1
echo "write done"
python $R/scripts/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_fetch_write.csv > /dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
python $R/bench.py --config synthetic_720p --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_synthetic_720p.json 2>> $O/bench.err
python $R/bench.py --config euroc_mh03_rd --no-cpu-baseline --steps 50 --warmup 5 > $O/bench_euroc_mh03_rd.json 2>> $O/bench.err
echo "all done"
