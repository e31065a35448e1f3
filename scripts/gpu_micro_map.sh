R=$GRAFT_REPO_ROOT
for n in base aplain; do for m in 0 1 2; do for S in 28 39; do
  echo "== $n map $m split $S"
  OG_MAP=$m timeout -k 10 60 $R/scripts/micro/bin/og_$n 715 48 33288 $S | tail -n 2 | cut -c1-110
done; done; done
# the pruned shape of configs[2]: 13 tiles ~ 715 x 256 ... (Nt = 2)
for n in base aplain; do for m in 0 1 2; do
  echo "== $n map $m, 32 operations (Nt = 2), split 39"
  OG_MAP=$m timeout -k 10 60 $R/scripts/micro/bin/og_$n 715 32 33288 39 | tail -n 2 | cut -c1-110
done; done
