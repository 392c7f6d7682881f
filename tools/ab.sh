for v in "$@"; do
  PT_HIP_LIB=$GRAFT_REPO_ROOT/path-tracing_amd/lib/libpt_$v.so python bench.py --cpu-seconds 0 --steps 3 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', round(j['value'],1), round(j['roofline']['kernel_ms'],2))"
done
