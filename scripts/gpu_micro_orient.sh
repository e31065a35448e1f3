# orbit GEMM micro-benchmark variants (see the header of scripts/micro/orbit_gemm.hip): wave priority raised over the MFMA block
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "-DORIENT4 -DWR=60" "-DORIENT4 -DWR=60 -DSETPRIO=1" "-DORIENT4 -DWR=60 -DSETPRIO=3" "-DORIENT4 -DWR=60" "-DORIENT4 -DWR=60 -DSETPRIO=1"; do
  hipcc --offload-arch=gfx950 -O3 -DNWM=2 -DNWN=2 -DTK=16 $v -o /tmp/og $R/scripts/micro/orbit_gemm.hip 2>/dev/null || { echo "compile failed: $v"; continue; }
  echo "== $v"
  for S in 28; do timeout -k 10 120 /tmp/og 715 48 33288 $S | tail -n 2; timeout -k 10 120 /tmp/og 715 48 33288 $S | tail -n 2 | head -n 1; done
done
