python -m pytest tests/test_hip_model.py -x -q -m gpu -k "scale_invariant_loss_inside" 2>&1 | tail -5
