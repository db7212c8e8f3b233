timeout 900 python -m pytest tests/test_gpu_tile_search.py -m gpu -x -q 2>&1 | tail -5
