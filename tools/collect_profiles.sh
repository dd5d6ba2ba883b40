#!/bin/bash
# collects the per-round profile set on the GPU box (run through gpurun from the repo root): kernel stats + shapes for the
# fp32 headline and the bf16 mode, PMC FETCH/WRITE passes (separate runs), unprofiled bench lines.  usage: tools/collect_profiles.sh <tag>
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-wgrad-overlap"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32 -o k -- python3 $R/bench.py $COMMON --kernel-report $O/fp32_shapes.json > $O/fp32_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -o k -- python3 $R/bench.py $COMMON --dtype bf16 --kernel-report $O/bf16_shapes.json > $O/bf16_prof.log 2>&1
PM="--steps 3 --warmup 1 --no-infer --no-cpu-baseline --no-wgrad-overlap"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bf16_fetch -o p -- python3 $R/bench.py $PM --dtype bf16 > $O/bf16_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bf16_write -o p -- python3 $R/bench.py $PM --dtype bf16 > $O/bf16_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fp32_fetch -o p -- python3 $R/bench.py $PM > $O/fp32_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fp32_write -o p -- python3 $R/bench.py $PM > $O/fp32_write.log 2>&1
cd $R
python3 bench.py > $O/fp32_bench_line.log 2>&1
python3 bench.py --dtype bf16 --graph > $O/bf16_bench_line.log 2>&1
python3 bench.py --no-wgrad-overlap --no-infer --no-cpu-baseline > $O/fp32_bench_line_no_overlap.log 2>&1
echo done
