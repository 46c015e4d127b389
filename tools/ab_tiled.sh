cd $GRAFT_REPO_ROOT
for rnd in 1 2; do for s in "X=1" "ISR_SPLIT_ALGO=0"; do
 out=$(env $s python3 bench.py --mode tiled --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.1f fps %.3f ms | %s' % (d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('conv3x3_','').replace('_kernel',''), v['ms_per_frame']) for k,v in d['roofline']['kernels'].items())))")
 echo "round $rnd [$s] $out"; done; done
