#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python tools/pipeline_soak.py 4000 2>&1 | tail -12
