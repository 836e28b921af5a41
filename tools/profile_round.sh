#!/bin/bash
# Run ON THE GPU BOX from the repo root: bench line, rocprofv3 kernel stats of the same command,
# and HBM-traffic counters (separate --pmc passes, as the microarch guide prescribes).
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/round; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu --no-power > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-power > $OUT/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/config5.py > $OUT/cal_fetch.log 2>&1; echo "cal fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- python3 $R/tools/config5.py > $OUT/cal_write.log 2>&1; echo "cal write rc=$?"
cd $R; ls $OUT
