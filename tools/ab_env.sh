# usage: bash tools/ab_env.sh VAR v1 v2 ...   (runs bench.py with VAR=v for each v)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --cpu-seconds 0 --steps 3 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$VAR=$v', round(j['value'],1), round(j['roofline']['kernel_ms'],2))"
done
