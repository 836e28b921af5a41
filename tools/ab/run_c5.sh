#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/ab/time_rfft.py 4096 2>&1 | tail -1
timeout -k 10 120 python tools/ab/time_rfft.py 8192 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-cpu --no-power --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v['frac'],3),round(v['avg_launch_ms'],3)) for k,v in d['roofline_rfft'].items()}, round(d['roofline_soft_demap']['frac'],3), round(d['value']/1e9,1))"
timeout -k 10 120 python tools/ab/time_rfft.py 4096 2>&1 | tail -1
