timeout 900 python -m pytest tests/test_gpu_batch_builder.py tests/test_gpu_train_cli.py -m gpu -x -q 2>&1 | tail -25 | cut -c1-400
