cd $GRAFT_REPO_ROOT
rm -f gpurun_out/x9_*.txt
python -m pytest tests -m gpu -x -q > gpurun_out/x9_tests.log 2>&1; tail -3 gpurun_out/x9_tests.log
python tools/fftbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/x9_fftbench.txt; python tools/fftbench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/x9_fftbench.txt
B="python bench.py --steps 300 --warmup 20 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants"
for i in 1 2 3; do $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/x9_bench.txt; done
