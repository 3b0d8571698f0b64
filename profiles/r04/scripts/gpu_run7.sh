#!/bin/bash
# r04 run 7: backtrace of the abort (rocgdb)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd "$R"
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGABRT stop" -ex run -ex "bt 40" --args python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "row_serial_batched_launch or per_pixel_bit_exact" > /tmp/gdb.log 2>&1
grep -v "New Thread\|exited\|RCCL\|warning:\|^$\|AMDGPU Wave" /tmp/gdb.log | tail -70
