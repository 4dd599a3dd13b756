#!/bin/bash
# rocprofv3 --kernel-trace --stats of the other bench modes (cfg-4: --pair text, cfg-5-shaped: --pair roundrobin --batch 128); summaries go to
# profiles/<round>_kernel_stats_text.csv / _roundrobin.csv (copy by hand from gpurun_out/prof_<round>_pairs/).  The program itself follows `--`.
set -e
R=${1:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_${R}_pairs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/text -o bench -- python3 $ROOT/bench.py --pair text --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_text.json 2> $OUT/text.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roundrobin -o bench -- python3 $ROOT/bench.py --pair roundrobin --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_roundrobin.json 2> $OUT/roundrobin.log
find $OUT -name "*kernel_stats.csv"
