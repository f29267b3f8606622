cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matmul_gpu.py -x -q -k "full_size or every_tile or golden or matches_oracle" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_shapes_gpu.py -x -q -k "llama_projection" 2>&1 | tail -3
tools/run_variants.sh "tools/gemm_clock.py 2048,128,1920 3072,896,128" nohook instr nohook instr > gpurun_out/r03_hook.txt 2>&1
tools/run_variants.sh "tools/gemm_clock.py K=14336 12288,1024,1024" nohook instr >> gpurun_out/r03_hook.txt 2>&1
grep -E "===|loop cycles" gpurun_out/r03_hook.txt
