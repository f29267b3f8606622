cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_matmul_gpu.py tests/test_stress_gpu.py tests/test_grouped_gpu.py tests/test_qlinear_gpu.py tests/test_decoder_chain_gpu.py tests/test_tp_gpu.py -x -q 2>&1 | tail -8
