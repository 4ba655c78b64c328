export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/scene_tr -o s -- python $GRAFT_REPO_ROOT/tools/scene_trace.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/scene_trace.py --report /tmp/scene_tr 12 > $GRAFT_REPO_ROOT/gpurun_out/r4e_scene_trace.txt 2>&1; head -45 $GRAFT_REPO_ROOT/gpurun_out/r4e_scene_trace.txt | cut -c1-250
