#!/bin/bash
# call 40: the final tree (label guard in the loss kernel, decoder refactor): the default bench line, then smoke()
mkdir -p gpurun_out/r06_c40
timeout 130 python bench.py > gpurun_out/r06_c40/bench.json 2> gpurun_out/r06_c40/bench.err
tail -c 600 gpurun_out/r06_c40/bench.json | head -c 400; echo
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_c40/smoke.txt 2>&1
tail -1 gpurun_out/r06_c40/smoke.txt
