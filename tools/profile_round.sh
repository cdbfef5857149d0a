#!/bin/bash
# tools/profile_round.sh <tag> <kernel substring> -- on the GPU box: bench JSON, rocprofv3 kernel-trace stats and the two PMC traffic passes of
# bench.py's default workload; raw output under gpurun_out/<tag>/, summaries into profiles/ (copy them back into the repo and commit).
tag=$1; kern=$2
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $root
python bench.py --no-cpu-baseline --no-other-configs > $out/bench.json 2> $out/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $out/pw.log 2>&1
mkdir -p $out/run/pmc_fetch $out/run/pmc_write
cp $(find $out/pf -name "*counter_collection.csv" | head -1) $out/run/pmc_fetch/
cp $(find $out/pw -name "*counter_collection.csv" | head -1) $out/run/pmc_write/
cp $out/bench.json $out/run/bench.json
mkdir -p $root/gpurun_out/profiles_$tag
(cd $root && python tools/pmc_summary.py $out/run $tag "$kern" && cp profiles/${tag}_pmc_traffic.json gpurun_out/profiles_$tag/)
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $root/gpurun_out/profiles_$tag/${tag}_wfm_kernel_stats.csv
cp $out/bench.json $root/gpurun_out/profiles_$tag/${tag}_bench_nocpu.json
head -8 $root/gpurun_out/profiles_$tag/${tag}_wfm_kernel_stats.csv | cut -c1-200
