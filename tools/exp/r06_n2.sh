python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k channels_last_vision 2>&1 | grep -E "Error|assert|gn" | head -20
