cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matmul_gpu.py -x -q -k "chained_segments" 2>&1 | tail -6
