for f in tests/test_gpu_api.py tests/test_gpu_boundary.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_eval.py tests/test_gpu_init.py tests/test_gpu_normal.py tests/test_cpp_api.py; do
  python -m pytest $f -m gpu -q > /tmp/out.txt 2>&1; rc=$?; echo "$f rc=$rc $(grep -E 'passed|failed' /tmp/out.txt | tail -1) $(grep -c 'Segmentation' /tmp/out.txt)"
done
