cd $GRAFT_REPO_ROOT
for n in 3 2 4 6 3; do
  echo -n "streams $n: "; SAVSR_STREAMS=$n timeout 600 python3 bench.py --clips-per-step $n --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'])"
done
echo -n "streams 3, 6 clips per step: "; SAVSR_STREAMS=3 timeout 600 python3 bench.py --clips-per-step 6 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
