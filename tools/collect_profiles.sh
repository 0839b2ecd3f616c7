#!/bin/bash
# GPU box (one gpurun call): every number and rocprof summary committed under profiles/ for this round.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- tools/collect_profiles.sh r03
set -o pipefail
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
export VNF_TUNE_CACHE=$OUT/tune_cache.txt
cd $ROOT
STEPS=6
if [ -z "$SKIP_EARLY" ]; then   # SKIP_EARLY=1: resume after the bf16 counter / trace passes
# 0. tune cache + the plan listing that keys the kernel trace by layer
VNF_PRINT_PLAN=1 python tools/pmc_one_step.py 256 3 > $OUT/plan.txt 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
# 1. counters, each set in its own pass (kernel trace only beside --pmc)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o f --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 4 > $OUT/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o w --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 4 > $OUT/pmc_w.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/pmc_m -o m --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 4 > $OUT/pmc_m.log 2>&1 || exit 1
# 2. per-layer kernel trace of the benchmarked (3-lane) mode
rocprofv3 --kernel-trace -d $OUT/kt_layers -o l --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 $STEPS > $OUT/kt_layers.log 2>&1 || exit 1
cd $ROOT
python tools/traffic_from_pmc.py $OUT/pmc_f/f_counter_collection.csv $OUT/pmc_w/w_counter_collection.csv 7 $OUT/${R}_traffic.json > /dev/null || exit 1
python tools/mfma_busy_from_pmc.py $OUT/pmc_m/m_counter_collection.csv $OUT/pmc_m/m_kernel_trace.csv > $OUT/${R}_mfma_busy_bs256_bf16.txt || exit 1
python tools/layer_trace.py $OUT/plan.txt $OUT/kt_layers/l_kernel_trace.csv $STEPS 3 > $OUT/${R}_irv1_bs256_bf16_3lane_layer_trace.txt || exit 1
fi
cd $ROOT
# 2b. the same per-layer table for the in-gate dtype (f16x2: planar split-f16, fused stem / Block35 / Block17 kernels)
VNF_PRINT_PLAN=1 python tools/pmc_one_step.py 256 3 f16x2 > $OUT/plan_f16x2.txt 2>/dev/null || exit 1
( cd /tmp && rocprofv3 --kernel-trace -d $OUT/kt_layers_x -o l --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 $STEPS f16x2 > $OUT/kt_layers_x.log 2>&1 ) || exit 1
python tools/layer_trace.py $OUT/plan_f16x2.txt $OUT/kt_layers_x/l_kernel_trace.csv $STEPS 3 > $OUT/${R}_irv1_bs256_f16x2_3lane_layer_trace.txt || exit 1
# 2c. where the waves of each kernel wait (SQ wait-state counters, one lane so the kernels do not overlap)
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_s -o s --output-format csv -- python3 $ROOT/tools/pmc_one_step.py 256 2 bf16 1 > $OUT/pmc_s.log 2>&1 ) || exit 1
python tools/pmc_table.py $OUT/pmc_s/s_counter_collection.csv 0.02 > $OUT/${R}_sq_wait_states_bf16.txt || exit 1
cp $OUT/${R}_traffic.json $ROOT/profiles/${R}_traffic.json   # bench.py quotes it in roofline.traffic
# 3. the bench lines
python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err || exit 1
python bench.py --workload detect --detectors 2 > $OUT/${R}_bench_detect.json 2>/dev/null || exit 1
python bench.py --workload embed --model ir100 --no-cpu-baseline > $OUT/${R}_bench_ir100.json 2>/dev/null || exit 1
python tools/run_layers.py bf16 2>/dev/null | grep -av amdgpu.ids > $OUT/${R}_irv1_bs256_bf16_layer_times.txt || exit 1   # one lane, no internal half-batch forks
python tools/run_layers.py f16x2 2>/dev/null | grep -av amdgpu.ids > $OUT/${R}_irv1_bs256_f16x2_layer_times.txt || exit 1
python bench.py --workload stream --dtype f16x2 --no-cpu-baseline > $OUT/${R}_bench_stream_f16x2.json 2>/dev/null || exit 1
python bench.py --workload stream --no-cpu-baseline > $OUT/${R}_bench_stream_bf16.json 2>/dev/null || exit 1   # bf16: compute no longer hides the PCIe ceiling
python bench.py --workload pipeline --detectors 1 --no-cpu-baseline > $OUT/${R}_bench_pipeline_1handle.json 2>/dev/null || exit 1   # the default line uses two handles
# 3b. detector stage / layer tables (MTCNN cascade on 16 x 1080p; RetinaFace swap-in detector)
python tools/mtcnn_layers.py > $OUT/${R}_mtcnn_stage_times.txt 2>&1 || exit 1
VNF_RETINA_LAYERS=1 python tools/retina_time.py 1080 1920 16 3 > $OUT/${R}_retina_1080p.txt 2>&1 || exit 1
# 4. kernel-trace summary of the same default bench command (embed legs + pipeline leg)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_bench -o b --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --legs "" > $OUT/kt_bench.log 2>&1 || exit 1
cp $OUT/kt_bench/b_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/pmc_m/*trace* $OUT/pmc_s $OUT/kt_layers $OUT/kt_layers_x $OUT/kt_bench/*trace*
ls $OUT
