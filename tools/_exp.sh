python -m pytest tests/test_hip_ops.py tests/test_hip_properties.py tests/test_hip_graph.py tests/test_cabi_direct.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
echo OLD; RAMNET_HIP_LIB=rpg_ramnet_amd/abl/lib_old.so python tools/latency_probe.py --reps 300 2>/dev/null | tail -3
echo NEW; python tools/latency_probe.py --reps 300 2>/dev/null | tail -3
done
