for v in 0 3; do python tools/raymarch_only.py ejecta256 480x270 20 $v 2>&1 | tail -1; python tools/raymarch_only.py ejecta256 1920x1080 12 $v 2>&1 | tail -1; done
python -m pytest tests/test_render_gpu.py -m gpu -x -q -k "parity_with_oracle" 2>&1 | tail -2
