cd $GRAFT_REPO_ROOT
rm -f gpurun_out/xa_*.txt
build_x/asm_probe2 > gpurun_out/xa_asm.txt 2>&1; cat gpurun_out/xa_asm.txt
python -m pytest tests/test_gpu_fft_path.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q > gpurun_out/xa_tests.log 2>&1; tail -3 gpurun_out/xa_tests.log
for l in autoencoder-fft_amd/libaefft.so build_x/libaefft_xPK0.so autoencoder-fft_amd/libaefft.so build_x/libaefft_xPK0.so; do
  AEFFT_LIB=$PWD/$l PROF=1 python tools/cfgstep.py p2 300 2>&1 | grep -v amdgpu.ids | grep -E "p2:|sgrad|opmse|chain|kspec|kgrad" >> gpurun_out/xa_p2.txt
done
