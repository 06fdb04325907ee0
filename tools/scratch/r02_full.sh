#!/bin/bash
# the whole GPU suite, then the share probe (sequential and pipelined) and the round's profiles
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_full; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/share_probe.py > $O/share_seq.log 2>&1; grep "share 1/" $O/share_seq.log
SHARE_PIPELINED=1 timeout -k 10 200 python tools/share_probe.py > $O/share_pipe.log 2>&1; grep "share 1/" $O/share_pipe.log
bash tools/r02_profiles.sh ${1:-d}
