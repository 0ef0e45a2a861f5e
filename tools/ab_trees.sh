#!/bin/bash
# tools/ab_trees.sh OLD_TREE [ROUNDS] : interleaved steady-state bench runs of THIS tree's library and of another checkout's
# (its own bench.py, its own library: an A/B across an ABI change), on ONE box.  Prints kernel_us of every form each line carries.
OLD=${1:-_r3ref}; ROUNDS=${2:-3}
ARGS=${AB_ARGS:---steps 400 --warmup 50 --no-cpu-baseline --no-extra-points --no-selfcheck --fused 0}
NEWFORMS=${AB_FORMS:-contract,contract_no_done_list,terminal_obs_only,single_pass}
for rep in $(seq $ROUNDS); do
  (cd $OLD && python bench.py $ARGS 2>/dev/null | python $OLDPWD/tools/ab_show.py OLD)
  python bench.py $ARGS --forms $NEWFORMS 2>/dev/null | python tools/ab_show.py NEW
done
