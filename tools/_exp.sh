python -m pytest tests/test_hip_ops.py tests/test_hip_model.py tests/test_cabi_direct.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
echo cache on; python tools/host_profile.py 2>/dev/null | head -1
echo cache off; RAMNET_DESC_CACHE=0 python tools/host_profile.py 2>/dev/null | head -1
done
