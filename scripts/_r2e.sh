mkdir -p gpurun_out/r2e
python -m pytest tests/test_parity_grad_gpu.py -x -q -m gpu -s > gpurun_out/r2e/parity.log 2>&1; grep -E "B=|passed|failed|Error|assert" gpurun_out/r2e/parity.log | head -40
python -m pytest tests/test_gemm_gpu.py tests/test_qenc_gpu.py tests/test_optim_gpu.py -x -q -m gpu > gpurun_out/r2e/t2.log 2>&1; tail -3 gpurun_out/r2e/t2.log
