cd $GRAFT_REPO_ROOT
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1
tail -5 gpurun_out/profile_r03.log
python tools/bench_shapes.py 1 16 128 256 512 2048 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_llama_shapes.txt
python tools/bench_tp_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_tp_configs.txt
tail -3 gpurun_out/r03_llama_shapes.txt gpurun_out/r03_tp_configs.txt
