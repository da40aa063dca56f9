O=$(mktemp -d)
for form in "--steps-per-graph 8" "--steps-per-graph 2" "--streams 1" "--steps-per-graph 8 --stats-batch 8192"; do
  echo "== $form"
  python multipath-nn_amd/train-nets cifar10-ac --synthetic --synthetic-size 50000 10000 --iters 2500 --log-every 2500 --out $O/nets $form 2>&1 | grep -a "wall\|co-training" | tr '\r' '\n' | grep -a -v "Iteration"
done
rm -rf $O
