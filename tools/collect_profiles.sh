#!/bin/bash
# collects the per-round profile set on the GPU box (run through gpurun from the repo root); every step appends to
# gpurun_out/prof_<tag>/progress.log.  usage: tools/collect_profiles.sh <tag> [fp32|all]
#   kernel stats + shapes (fp32 headline, optionally the bf16 mode), PMC FETCH / WRITE passes (separate runs), SQ counters of
#   the MFMA kernels, a single-stream inference kernel trace, the unprofiled bench line
set -e
TAG=$1
WHAT=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
say() { echo "$(date +%T) $*" >> $O/progress.log; }
COMMON="--steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-wgrad-overlap --step-mode graph"
PM="--steps 3 --warmup 1 --no-infer --no-cpu-baseline --no-wgrad-overlap --step-mode graph"
say "fp32 kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32 -o k -- python3 $R/bench.py $COMMON --kernel-report $O/fp32_shapes.json > $O/fp32_prof.log 2>&1
say "fp32 FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fp32_fetch -o p -- python3 $R/bench.py $PM > $O/fp32_fetch.log 2>&1
say "fp32 WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fp32_write -o p -- python3 $R/bench.py $PM > $O/fp32_write.log 2>&1
say "fp32 SQ counters"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/fp32_sq -o p -- python3 $R/bench.py $PM > $O/fp32_sq.log 2>&1
say "fp32 LDS counters"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/fp32_lds -o p -- python3 $R/bench.py $PM > $O/fp32_lds.log 2>&1
say "inference, single stream, kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer -o k -- python3 $R/tools/bench_infer.py 512,512,400 16 --single-stream > $O/infer_prof.log 2>&1
if [ "$WHAT" = "all" ]; then
say "bf16 kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -o k -- python3 $R/bench.py $COMMON --dtype bf16 --kernel-report $O/bf16_shapes.json > $O/bf16_prof.log 2>&1
say "bf16 FETCH / WRITE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bf16_fetch -o p -- python3 $R/bench.py $PM --dtype bf16 > $O/bf16_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bf16_write -o p -- python3 $R/bench.py $PM --dtype bf16 > $O/bf16_write.log 2>&1
fi
# the per-dispatch traces are not needed (statistics and counter tables are) and gpurun returns at most 64 MiB
find $O -name "*_kernel_trace.csv" -delete
cd $R
say "unprofiled bench lines"
python3 bench.py > $O/fp32_bench_line.log 2>&1
if [ "$WHAT" = "all" ]; then python3 bench.py --dtype bf16 --no-cpu-baseline > $O/bf16_bench_line.log 2>&1; fi
python3 bench.py --no-wgrad-overlap --no-infer --no-cpu-baseline > $O/fp32_bench_line_no_overlap.log 2>&1
say done
echo done
