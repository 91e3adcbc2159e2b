cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/diag_lockflow.py c1 2 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp"
HTM_FLOW_LOCK=0 timeout -k 10 300 python tools/diag_lockflow.py c1 2 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp"
