#!/bin/bash
# GPU call 11: tower conv1 diagnostic (matrix-core vs direct kernels vs float64)
python tools/r06/tower_diag.py 2>&1 | tail -12
