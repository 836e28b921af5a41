#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in s_base s_max-ilp s_max-memory-clause s_base s_max-ilp; do
  GF3_LIB=$PWD/tools/ab/$v.so python bench.py --no-cpu --no-config5 --no-stream --no-power 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e9,1), d['ber'], d['sync_exact'], round(d['roofline']['avg_launch_ms'],3), round(d['roofline_sync']['avg_launch_ms'],3))"
  GF3_LIB=$PWD/tools/ab/$v.so python tools/ab/time_screen.py 2>&1 | tail -1 | cut -c1-150
done
