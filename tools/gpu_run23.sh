#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/r02x_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02x_pytest.log
tail -12 gpurun_out/r02x_pytest.log
