# debugging aid: bench.py on one rank and on N ranks sharing the GPU (PMH_BENCH_TRANSPORT=host), same arguments; prints checksums and step mixes.  usage: two_rank_dbg.sh [N]
cd $GRAFT_REPO_ROOT
N=${1:-2}
run() {
  tag=$1; shift
  python bench.py "$@" --details gpurun_out/one_$tag.json > /dev/null 2> gpurun_out/one_$tag.err
  PMH_BENCH_TRANSPORT=host python bench.py --gpus $N "$@" --details gpurun_out/two_$tag.json > /dev/null 2> gpurun_out/two_$tag.err || tail -5 gpurun_out/two_$tag.err
  python3 - $tag <<'P'
import json,sys
t=sys.argv[1]
a=json.load(open("gpurun_out/one_%s.json"%t)); b=json.load(open("gpurun_out/two_%s.json"%t))
print(t, b["n_gpus"], a["config"]["checksum"], b["config"]["checksum"])
print("  ", a["config"]["steps_by_type"]); print("  ", b["config"]["steps_by_type"])
P
}
run orbit --nel 9 --steps 40 --warmup 2 --no-cpu-baseline --no-c2 --no-iterative
run iter --nel 7 --steps 12 --warmup 2 --no-cpu-baseline --no-c2 --kplus iterative --no-iterative
run c3 --sub 4,4,4 --nel 5 --dense-coarse --steps 30 --warmup 2 --no-cpu-baseline --no-c2 --no-iterative
run svm --workload svm --svm-n 200000 --steps 10 --warmup 2
