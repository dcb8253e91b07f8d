#!/bin/bash
# GPU box: rocprofv3 summaries and bench lines for profiles/ (tag = $1).  Output under gpurun_out/<tag>/.
# Order: kernel trace, the two PMC passes (condensed into profiles/<tag>_pmc_fetch_write.csv, which bench.py reads for
# roofline.traffic), then the bench lines.
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --end-to-end-frames 0 --sequences 0 > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --no-cpu-baseline --end-to-end-frames 0 --sequences 0 --steps 30 --warmup 5 > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --no-cpu-baseline --end-to-end-frames 0 --sequences 0 --steps 30 --warmup 5 > $O/pmc_write.log 2>&1
echo "write done"
python $R/scripts/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_all.csv > /dev/null
grep -v "^__amd\|^at::" $O/pmc_all.csv > $O/pmc_fetch_write.csv
cp $O/pmc_fetch_write.csv $R/profiles/${TAG}_pmc_fetch_write.csv
# bench.py reads profiles/<round>_pmc_fetch_write.csv (+ .meta.json): refresh it in place so that the bench lines below carry `traffic`
ROUND=${TAG%%_*}
cp $O/pmc_fetch_write.csv $R/profiles/${ROUND}_pmc_fetch_write.csv
python -c "import sys, json; sys.path.insert(0, '$R'); import bench; json.dump({'kernel_source_sha256': bench.kernel_source_hash(), 'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --no-cpu-baseline --end-to-end-frames 0 --sequences 0 --steps 30 --warmup 5'}, open('$O/pmc_fetch_write.meta.json', 'w'))"
cp $O/pmc_fetch_write.meta.json $R/profiles/${TAG}_pmc_fetch_write.meta.json
cp $O/pmc_fetch_write.meta.json $R/profiles/${ROUND}_pmc_fetch_write.meta.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_all.csv
python $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
# configs 3 and 5 WITH the CPU legs: feature_indices_identical / ate_rmse_gpu_vs_cpu_path_mm exist for every config
python $R/bench.py --config synthetic_720p --steps 30 --warmup 5 > $O/bench_synthetic_720p.json 2>> $O/bench.err
echo "720p done"
python $R/bench.py --config euroc_mh03_rd --steps 50 --warmup 5 > $O/bench_euroc_mh03_rd.json 2>> $O/bench.err
echo "mh03 done"
python $R/bench.py --serial --no-cpu-baseline --end-to-end-frames 0 --sequences 0 > $O/bench_serial.json 2>> $O/bench.err
python $R/bench.py --no-cpu-baseline --sequences 0 --end-to-end-frames 400 --steps 50 --warmup 5 > $O/bench_long_400_frames.json 2>> $O/bench.err
echo "all done"
