#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
