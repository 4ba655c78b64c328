export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout ${SUITE_TIMEOUT:-2300} python -m pytest tests -q -m gpu -rf --durations=80 ${PYTEST_EXTRA} > $O/suite.log 2>&1; tail -120 $O/suite.log
