cd $GRAFT_REPO_ROOT
python -m pytest tests/test_pipeline.py -x -q 2>&1 | tail -3
