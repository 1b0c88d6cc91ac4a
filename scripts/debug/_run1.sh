cd $GRAFT_REPO_ROOT
for m in late early late early; do
echo -n "split defer $m: "; MISEG_SPLIT_DEFER=$m timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['exchange_check'], d['replay_check']['grad_rel_err'])"
done
