#!/bin/bash
# rocprofv3 artefacts of a round (GPU box; writes under gpurun_out/prof_$1, summaries are then copied into profiles/ by tools/profile_summarise.py).
#   pass 1  --kernel-trace --stats        of the default bench.py run (the same command line the driver uses, shorter)
#   pass 2/3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes: TCC has 4 slots, FETCH_SIZE costs 3) of the dominant kernel's launch
#   pass 4  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of the ESM-2-150M encoder forward (north_star: MFMA utilisation of the encoder forward)
# PMC passes never combine with --sys-trace etc. (gpurun refuses that); the program itself follows `--` (no env / bash -c hop).
set -e
R=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_profiled.json 2> $OUT/stats.log
for what in ffn1 ffn1fwd; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$what -o p -- python3 $ROOT/tools/gemm_only.py -1 $what 3 > /dev/null 2> $OUT/fetch_$what.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$what -o p -- python3 $ROOT/tools/gemm_only.py -1 $what 3 > /dev/null 2> $OUT/write_$what.log
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_fwd -o p -- python3 $ROOT/tools/encoder_fwd_only.py 2 > /dev/null 2> $OUT/mfma_fwd.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_step -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/mfma_step.log
find $OUT -name "*.csv" | head -40
