cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
