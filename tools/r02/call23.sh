cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -25
