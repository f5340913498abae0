cd $GRAFT_REPO_ROOT
( time python tools/extra_fuzz.py 18000 18500 ) > gpurun_out/fuzz_18000.txt 2>&1
tail -5 gpurun_out/fuzz_18000.txt
