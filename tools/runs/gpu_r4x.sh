#!/bin/bash
set -u
timeout 100 python -m pytest tests/test_hip_optim.py tests/test_hip_layer.py -m gpu -q -p no:cacheprovider -k "step_cached or full_model_train_step or train_epoch_mirrors" 2>&1 | tail -2
