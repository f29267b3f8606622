cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_matmul_gpu.py tests/test_stress_gpu.py tests/test_qlinear_gpu.py tests/test_graph_gpu.py tests/test_tp_gpu.py tests/test_decoder_chain_gpu.py tests/test_grouped_gpu.py tests/test_cabi_cpp_gpu.py -x -q 2>&1 | tail -4
for s in "128 1024" "256 1024" "512 1024" "128 4096" "128 5120 2560,128,2432"; do python tools/split_clock.py $s 2>&1 | grep -v amdgpu.ids | grep -E "mm::|kernel"; done
