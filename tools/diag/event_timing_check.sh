cd $GRAFT_REPO_ROOT
for args in "" "--no-overlap" "--steps 3" "--no-overlap --steps 3"; do echo "== $args"; python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['achieved'], d['lm_step']['gemm_kinds']['gate_up_swiglu'])"; done
