set -e
timeout -k 10 900 python -m pytest tests/test_gpu_apply.py -x -q -k "fused_pass_forms or every_mixed_radix or sizes_that_are_not or random_shapes or mixed_radix_even" > gpurun_out/r03_alt3_tests.log 2>&1; tail -3 gpurun_out/r03_alt3_tests.log
timeout -k 10 600 python tools/prof_sizes.py 192 384 2>&1 | grep "mixed radix" > gpurun_out/r03_alt3_sizes.log; cat gpurun_out/r03_alt3_sizes.log
