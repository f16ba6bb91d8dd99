cd $GRAFT_REPO_ROOT
rm -f gpurun_out/x8_*.txt
build_x/asm_probe > gpurun_out/x8_asm.txt 2>&1; cat gpurun_out/x8_asm.txt
python -m pytest tests/test_gpu_fft_path.py tests/test_gpu_configs.py tests/test_gpu_round2.py -x -q > gpurun_out/x8_tests.log 2>&1; tail -3 gpurun_out/x8_tests.log
python tools/fftbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/x8_fftbench.txt; python tools/fftbench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/x8_fftbench.txt
B="python bench.py --steps 300 --warmup 20 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants"
for i in 1 2 3; do $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/x8_bench.txt; done
