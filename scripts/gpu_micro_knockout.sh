# orbit GEMM micro-benchmark: knock-out variants (timing only for the NO* ones) -- where do the idle matrix-pipe cycles come from?
R=$GRAFT_REPO_ROOT
for n in ${VARIANTS:-base gfirst defsign gfirst_defsign nogather noaload nobar nostore nomem nomem_nobar base gfirst}; do
  echo "== $n"
  timeout -k 10 60 $R/scripts/micro/bin/og_$n 715 48 33288 28 | tail -n 2
done
timeout -k 10 60 $R/scripts/micro/bin/mfma_f64 | head -8
