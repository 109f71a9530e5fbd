timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_gpu.py tests/test_gpu_scale.py tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python scripts/pcie_rate.py 2>/dev/null
BATCH=16 timeout 300 python scripts/pcie_rate.py 2>/dev/null
BATCH=256 NF=512 timeout 300 python scripts/pcie_rate.py 2>/dev/null
