cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout -k 10 240 python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run --lm-priority
run --lm-priority --tile-dma 5
run --lm-priority --tile-dma 4
run --lm-priority --tile-dma 0
