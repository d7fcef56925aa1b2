mkdir -p gpurun_out
timeout 600 python tools/eval_split_bf16.py > gpurun_out/r06_split_bf16_eval.txt 2>&1
grep -v amdgpu gpurun_out/r06_split_bf16_eval.txt
