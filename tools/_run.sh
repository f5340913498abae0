cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_icp.py tests/test_tracking.py tests/test_fuzz_parity.py -k "icp or track" -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do python tools/dbg_icp.py 2>&1 | tail -1; done
for i in 1 2; do python tools/dbg_tracking.py 2>&1 | grep "ms per frame"; done
