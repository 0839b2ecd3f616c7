#!/bin/bash
# GPU box: run the encoder parity tests once per forced tile configuration (ids given as arguments)
for c in "$@"; do
  echo "== cfg $c"
  VNF_AUTOTUNE=0 VNF_FORCE_CFG=$c timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -3 || exit 1
done
