#!/bin/bash
# tools/profile_bench.sh <tag> <kernel substring> <bench script> [bench args...] -- on the GPU box: rocprofv3 kernel-trace stats and the two PMC
# traffic passes (FETCH_SIZE / WRITE_SIZE, each in its own run, counters + --kernel-trace only) of one bench script's workload; raw output under
# gpurun_out/<tag>/, summaries into gpurun_out/profiles_<tag>/ (copy them into profiles/ and commit).
tag=$1; kern=$2; script=$3; shift 3
[ "$script" = "bench.py" ] && set -- "$@" --no-other-configs      # the driver line's extra legs are child processes of their own: not part of this profile
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
out=$root/gpurun_out/$tag; mkdir -p $out $root/gpurun_out/profiles_$tag
cd /tmp && export TMPDIR=/tmp; cd $root
python $script "$@" --no-cpu-baseline > $out/bench.json 2> $out/bench.err      # the script's default step count: the same warm state as the trace below
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $script "$@" --steps 120 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -- python $script "$@" --steps 2 --warmup 1 --no-cpu-baseline > $out/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -- python $script "$@" --steps 2 --warmup 1 --no-cpu-baseline > $out/pw.log 2>&1
mkdir -p $out/run/pmc_fetch $out/run/pmc_write
cp $(find $out/pf -name "*counter_collection.csv" | head -1) $out/run/pmc_fetch/ 2>/dev/null
cp $(find $out/pw -name "*counter_collection.csv" | head -1) $out/run/pmc_write/ 2>/dev/null
tail -1 $out/bench.json > $out/run/bench.json
(cd $root && python tools/pmc_summary.py $out/run $tag "$kern" "$script $*" && cp profiles/${tag}_pmc_traffic.json gpurun_out/profiles_$tag/)
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $root/gpurun_out/profiles_$tag/${tag}_kernel_stats.csv "$tag: rocprofv3 --kernel-trace --stats -- python $script $* --steps 120 --warmup 5 (durations in ns)"
cp $out/run/bench.json $root/gpurun_out/profiles_$tag/${tag}_bench_nocpu.json
head -8 $root/gpurun_out/profiles_$tag/${tag}_kernel_stats.csv | cut -c1-220
