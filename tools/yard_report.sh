#!/bin/bash
# HIP error vs the reference's own float32 deviation for every yardstick comparison (tests/conftest.py Yardstick)
HECKA_YARD_REPORT=1 timeout -k 10 400 python -u -m pytest tests -m gpu -q -s --timeout 240 --timeout-method thread -k "train_step or svgd or learnable_hyper or hypernetwork_step or model_kernel_matches" 2>&1 | grep "YARD\|passed\|failed\|Error" > gpurun_out/yard_report.txt
grep -c YARD gpurun_out/yard_report.txt; grep "FAIL\|passed\|failed" gpurun_out/yard_report.txt | head -40
