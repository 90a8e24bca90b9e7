#!/bin/bash
python -m pytest tests/test_match_state_gpu.py -x -q -m gpu 2>&1 | tail -3
