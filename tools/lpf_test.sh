for l in 8 12; do CCAL_GRAMV_LPF=$l python -m pytest tests/test_gpu_normal.py -m gpu -q -k "all_lane_mappings or frames_of_any_size or build_normal_matches" 2>&1 | grep -E "passed|failed"; done
