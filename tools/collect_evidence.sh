#!/bin/bash
# Round evidence on the GPU box (from the repo root): the driver's bench command, rocprofv3 kernel tables of every benchmarked workload,
# and the HBM-traffic PMC passes of the dominant NS kernel.  usage: bash tools/collect_evidence.sh r06   -> gpurun_out/<tag>_*
set -u
TAG=${1:-r06}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_1gpu.json 2> $O/${TAG}_bench_1gpu.err
echo "bench rc $?"
prof() {  # name, command...
  name=$1; shift
  rm -rf /tmp/pf_$name
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -o $name -- "$@" > /tmp/pf_$name.log 2>&1 )
  db=$(find /tmp/pf_$name -name "*.db" | head -1)
  python tools/prof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- $*" > $O/${TAG}_${name}_kernel_stats.txt
  echo "profile $name: $(wc -l < $O/${TAG}_${name}_kernel_stats.txt) lines"
}
cd $R
prof bench_nb80 python $R/bench.py --nb 80 --steps 10 --warmup 2 --no-extra-configs --no-cpu-baseline
prof oisst_nb300_groups3 python $R/tools/bench_oisst.py 300
prof oisst_nb38 python $R/tools/bench_oisst.py 38
prof synth512_nb4_fp16 python $R/tools/bench_synth512.py 4 fp16 rollout
DYF_TRAIN_OPERANDS=bf16 prof train_step_ns_b32_16bit python $R/tools/bench_train_step.py 32
DYF_TRAIN_OPERANDS=bf16 prof train_step_resnet_b64_16bit python $R/tools/bench_train_step_resnet.py 64
# HBM-side traffic of the dominant kernel (separate --pmc passes; MI355X_MICROARCH.md HBM section)
DYF_PMC_REGEX=conv_halo_rows bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_ns tools/bench_layers.py 80 3
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_ns > $O/${TAG}_rows_pmc_traffic_nb80.txt
rm -rf gpurun_out/${TAG}_pmc_ns
ls -la $O | grep $TAG
