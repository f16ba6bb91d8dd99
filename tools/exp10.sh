cd $GRAFT_REPO_ROOT
rm -f gpurun_out/xb_*.txt
python -m pytest tests/test_gpu_fft_path.py tests/test_gpu_configs.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q > gpurun_out/xb_tests.log 2>&1; tail -3 gpurun_out/xb_tests.log
python tools/fftbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/xb_fftbench.txt; python tools/fftbench.py 1024 96 2 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/xb_fftbench.txt
B="python bench.py --steps 300 --warmup 20 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants"
for i in 1 2 3; do $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/xb_bench.txt; done
bash tools/tl.sh xb_tl > /dev/null 2>&1; head -12 gpurun_out/xb_tl_tl.txt >> gpurun_out/xb_bench.txt
