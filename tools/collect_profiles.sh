#!/bin/bash
# GPU box (one gpurun call): every number and rocprof summary committed under profiles/ for this round.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- tools/collect_profiles.sh r01
set -o pipefail
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
export VNF_TUNE_CACHE=$OUT/tune_cache.txt
cd $ROOT
# 0. tune cache
python tools/pmc_one_step.py 256 1 > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
# 1. HBM traffic: counters in their own passes, kernel trace only
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 4 > $OUT/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 4 > $OUT/pmc_w.log 2>&1 || exit 1
cd $ROOT
python tools/traffic_from_pmc.py $OUT/pmc_f/f_counter_collection.csv $OUT/pmc_w/w_counter_collection.csv 4 $OUT/${R}_traffic.json > /dev/null || exit 1
cp $OUT/${R}_traffic.json $ROOT/profiles/${R}_traffic.json   # bench.py quotes it in roofline.traffic
# 2. the bench lines
python bench.py > $OUT/${R}_bench_embed.json 2> $OUT/bench_embed.err || exit 1
python bench.py --workload pipeline > $OUT/${R}_bench_pipeline.json 2> $OUT/bench_pipeline.err || exit 1
python bench.py --workload detect --detectors 2 > $OUT/${R}_bench_detect.json 2>/dev/null || exit 1
python bench.py --dtype f32 --no-cpu-baseline > $OUT/${R}_bench_embed_f32.json 2>/dev/null || exit 1
python bench.py --model ir100 --no-cpu-baseline > $OUT/${R}_bench_ir100.json 2>/dev/null || exit 1
python tools/profile_encoder.py 256 bf16 > $OUT/${R}_irv1_bs256_bf16_layer_times.txt 2>/dev/null || exit 1
# 3. kernel-trace summaries of the same bench commands
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_embed -o e --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/kt_embed.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/kt_pipe -o p --output-format csv -- python3 $ROOT/bench.py --workload pipeline --no-cpu-baseline > $OUT/kt_pipe.log 2>&1 || exit 1
cp $OUT/kt_embed/e_kernel_stats.csv $OUT/${R}_bench_embed_bs256_bf16_kernel_stats.csv
cp $OUT/kt_pipe/p_kernel_stats.csv $OUT/${R}_pipeline_1080p_kernel_stats.csv
rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/kt_embed/*trace* $OUT/kt_pipe/*trace*
ls $OUT
