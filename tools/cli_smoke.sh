# usage: bash tools/cli_smoke.sh -- the reference's CLI (run_skrec.py) for every in-scope model on the tiny golden data set
set -e
D=$(mktemp -d)
python - <<PY
import numpy as np, os
d = np.load("tests/golden/tiny_dataset.npz")
os.makedirs("$D/tiny", exist_ok=True)
for name in ("train", "test"):
    a = d[name]
    with open("$D/tiny/tiny." + name, "w") as f:
        for u, i, t in a:
            f.write(f"{u}\t{i}\t1.0\t{t}\n")
PY
cd $D
for m in BPRMF LightGCN LayerGCN GRU4RecPlus; do
  extra=""; [ $m = GRU4RecPlus ] && extra="--batch_size 16 --n_sample 64"
  PYTHONPATH=$GRAFT_REPO_ROOT/scikit-recommender_amd python $GRAFT_REPO_ROOT/scikit-recommender_amd/run_skrec.py --recommender $m --data_dir $D/tiny \
     --file_column UIRT --sep "\t" --epochs 2 --metric '["Recall","NDCG"]' --top_k '[5,10]' $extra 2>&1 | grep -E "best:|Error|Traceback" | sed "s/^/$m /"
done
