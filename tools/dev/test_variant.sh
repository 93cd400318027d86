#!/bin/bash
# run the field-kernel GPU tests against a variant library: bash tools/dev/test_variant.sh <name> [pytest -k expression]
cd "$(dirname "$0")/../.."
K=${2:-"image_walk or full_size_properties"}
timeout 300 python - "$1" "$K" <<'PY'
import importlib, sys, os
sys.path.insert(0, os.getcwd())
importlib.import_module('3dgp_amd')._lib.LIB_PATH = os.path.abspath(f'tools/dev/variants/{sys.argv[1]}.so')
import pytest
sys.exit(pytest.main(['tests/test_gpu_parity.py', '-q', '-x', '-k', sys.argv[2]]))
PY
