#!/bin/bash
# Wall time of ONE experiment run end to end: train-nets cifar10-ac --synthetic (CIFAR-10-sized sets: the statistics pass
# reads 50 000 + 10 000 images per net and log point) --iters N --log-every N, net after net against the default (the 8 nets
# co-trained, K-step joint hipGraph replays).   bash tools/experiment_wall.sh [iters] [log-every]  ->  stdout (profiles/r06_experiment_wall.txt)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-2500}
L=${2:-$N}
O=$(mktemp -d)
cd $R
for form in "--co-train 1" ""; do
  echo "== train-nets cifar10-ac --synthetic --synthetic-size 50000 10000 --iters $N --log-every $L $form"
  T0=$(date +%s.%N)
  python multipath-nn_amd/train-nets cifar10-ac --synthetic --synthetic-size 50000 10000 --iters $N --log-every $L --out $O/nets $form 2>&1 | grep -a "wall\|co-training\|Error\|error" | tr '\r' '\n' | grep -a -v "Iteration"
  echo "  process wall $(python3 -c "print('%.2f' % ($(date +%s.%N) - $T0))") s (start-up, compilation of nothing, dataset upload included)"
done
rm -rf $O
