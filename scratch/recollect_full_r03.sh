set -o pipefail
R=$GRAFT_REPO_ROOT
for w in "train_full 512" "train_full 4096" "train_prop 512"; do
  set -- $w
  bash $R/scratch/profile_r03.sh $1 $2 all || exit 1
done
echo profiled
