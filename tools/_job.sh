cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matmul_gpu.py tests/test_tp_gpu.py tests/test_tp_rccl_gpu.py -x -q -k "fp32_output or sharded_sum or rccl_worker" 2>&1 | tail -12
