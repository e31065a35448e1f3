# debugging aid: bench.py on one rank and on two ranks sharing the GPU (PMH_BENCH_TRANSPORT=host), same arguments; prints checksums and step mixes
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  python bench.py "$@" --details gpurun_out/one_$tag.json > /dev/null 2> gpurun_out/one_$tag.err
  PMH_BENCH_TRANSPORT=host python bench.py --gpus 2 "$@" --details gpurun_out/two_$tag.json > /dev/null 2> gpurun_out/two_$tag.err
  python3 - $tag <<'P'
import json,sys
t=sys.argv[1]
a=json.load(open("gpurun_out/one_%s.json"%t)); b=json.load(open("gpurun_out/two_%s.json"%t))
print(t, a["config"]["checksum"], b["config"]["checksum"])
print("  ", a["config"]["steps_by_type"]); print("  ", b["config"]["steps_by_type"])
P
}
run iter --nel 7 --steps 12 --warmup 2 --no-cpu-baseline --no-c2 --kplus iterative --no-iterative
run svm --workload svm --svm-n 200000 --steps 10 --warmup 2
