mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_rcnn_step_gpu.py tests/test_fullsize_gpu.py tests/test_fcos_step_gpu.py tests/test_conv_bf16_gpu.py -q -m gpu -k "rcnn_full_semisup or reference_trainer_golden or fullsize or amp or rounding_oracle or mirror or hipgraph or premasked or bit_planes or lanes or handoff or reduces_loss or trainable_stem or tight" > gpurun_out/r06_probe3_tests.txt 2>&1
tail -40 gpurun_out/r06_probe3_tests.txt
