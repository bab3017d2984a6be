cd /root/repo
python -m pytest tests/test_gpu_bottom_right.py -x -q 2>&1 | tail -15
for i in 1 2; do
AULE_LIBRARY_PATH=/root/repo/build/libaule_base.so python tools/ab_bench.py
python tools/ab_bench.py br
done
