for spec in "--chains-per-gpu 1024" "--chains-per-gpu 2048" "--chains-per-gpu 3072" "--chains-per-gpu 5000" "--chains-per-gpu 4096 --dim 200" "--chains-per-gpu 4096 --multitry 1" "--chains-per-gpu 4096 --dim 256"; do
for L in 0 19; do
python bench.py $spec --adapt --adapt-lag $L --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-dense --no-lag0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        j=json.loads(l); print('$spec', $L, 'burnin', j.get('burnin_value'), 'post', j['value'], j['config'].get('kernel'))
"
done; done
