cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/plane_read.hip -o /tmp/plane_read 2>/dev/null && /tmp/plane_read
