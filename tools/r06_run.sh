export PYTHONPATH=$PWD TMPDIR=/tmp
(time timeout 2400 python -m pytest tests -q -m gpu -x) 2>&1 | tail -8
