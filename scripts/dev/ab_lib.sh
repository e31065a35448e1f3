# A/B of two builds of the library on ONE box: permon_amd/libpermonhip_old.so against the tree's, the inner-Krylov bench on the driver's window, alternating, 3 rounds
cd $GRAFT_REPO_ROOT
cp permon_amd/libpermonhip.so /tmp/new.so; cp permon_amd/libpermonhip_old.so /tmp/old.so
for r in 1 2 3; do for v in old new; do
  cp /tmp/$v.so permon_amd/libpermonhip.so
  timeout -k 10 200 python bench.py --kplus iterative --steps 20 --warmup 5 --no-cpu-baseline --no-c2 --no-iterative > /tmp/ab.json 2> /tmp/ab.err
  python -c "
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))"
done; done
cp /tmp/new.so permon_amd/libpermonhip.so
