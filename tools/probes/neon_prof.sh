cd $GRAFT_REPO_ROOT
tools/kt.sh r05_neon_train_dense python tools/prof_neon.py --train --dense
tools/kt.sh r05_neon_train_plain python tools/prof_neon.py --train
tools/kt.sh r05_neon_infer_dense python tools/prof_neon.py --dense
tools/kt.sh r05_neon_infer_plain python tools/prof_neon.py
head -40 gpurun_out/r05_neon_train_dense.txt | cut -c1-170
