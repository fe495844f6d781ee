set -u
cd $GRAFT_REPO_ROOT
MGF_F6_OPTS=resort_every=0 python tools/flow_trace.py 64 40 6 2>&1 | tail -22
echo ---- settled
MGF_F6_OPTS=resort_every=0 python tools/flow_trace.py 64 420 6 2>&1 | tail -22
