cd $GRAFT_REPO_ROOT
TSDF_RAY_DBG=1 python tools/dbg_ray_only.py 3 2>&1 | grep "long waves" | tail -4
