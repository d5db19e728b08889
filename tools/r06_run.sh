export PYTHONPATH=$PWD TMPDIR=/tmp
(time timeout 2400 python -m pytest tests -q -m gpu) 2>&1 | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
