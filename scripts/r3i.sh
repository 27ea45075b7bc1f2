source scripts/r3_run.sh r3i
step hyp 600 python -m pytest tests/test_model_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "hypothesis or inference_fused or translator"
step fp8prof 900 bash scripts/r3_prof_fp8.sh r3i
tail -6 gpurun_out/r3i_hyp.log; cat gpurun_out/r3i_fp8_kernel_pmc.json
