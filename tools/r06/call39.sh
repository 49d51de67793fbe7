#!/bin/bash
# call 39: the decoder refactor (cell parameters travel with the call) on the device: decoder tests, the sgcls step, smoke
mkdir -p gpurun_out/r06_c39
timeout 100 python -m pytest tests/test_gpu_ops.py -q -x -k "decoder" > gpurun_out/r06_c39/decoder.txt 2>&1
grep -E "passed|failed" gpurun_out/r06_c39/decoder.txt | tail -1
timeout 90 python -m pytest tests/test_gpu_model.py -q -x -k "sgcls_train_step" > gpurun_out/r06_c39/model.txt 2>&1
grep -E "passed|failed" gpurun_out/r06_c39/model.txt | tail -1
