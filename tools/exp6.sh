cd $GRAFT_REPO_ROOT
rm -f gpurun_out/x6_*.txt
python -m pytest tests -m gpu -x -q > gpurun_out/x6_tests.log 2>&1; tail -5 gpurun_out/x6_tests.log
for l in autoencoder-fft_amd/libaefft.so build_x/libaefft_xGD0.so; do AEFFT_LIB=$PWD/$l PROF=1 python tools/cfgstep.py cfg5 5 2>&1 | grep -v amdgpu.ids | head -8 >> gpurun_out/x6_cfg5.txt; done
