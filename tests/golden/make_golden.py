"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden.py all

Each section runs in a fresh subprocess because the reference's sampler stream is a process-global
``std::mt19937 _gen(2020)`` (randint.h:20) with no re-seed API: "fresh process" is part of every
vector's definition.  Only data (inputs + outputs) is written; see ref_harness.py for how the
reference is imported.  The fixtures are small (.npz, a few hundred KB in total) and committed;
this script is committed with them so they can be regenerated and audited.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SCRATCH = os.path.join(REPO, "oracle", "_scratch")
DATA_DIR = os.path.join(SCRATCH, "tiny")
SEED = 2021  # run_skrec.py:56 default


# ------------------------------------------------------------------------------------------------
def make_dataset():
    """Our own seeded tiny implicit-feedback set (not reference data): 64 users x 96 items."""
    rng = np.random.default_rng(20260101)
    U, I = 64, 96
    pop = 1.0 / np.arange(1, I) ** 0.8  # item 95 never appears in train
    pop /= pop.sum()
    tr, te = [], []
    t = 0
    for u in range(U):
        n = int(rng.integers(5, 29))
        items = rng.choice(I - 1, size=n, replace=False, p=pop)
        n_te = max(1, n // 5)
        if u == 63:       # test-only user (no train row, zero-degree node)
            tr_items, te_items = [], items[:3]
        elif u == 5:      # train-only user (never evaluated)
            tr_items, te_items = items, []
        else:
            tr_items, te_items = items[:-n_te], items[-n_te:]
        for i in tr_items:
            tr.append((u, int(i), 1.0, t)); t += 1
        for i in te_items:
            te.append((u, int(i), 1.0, t)); t += 1
    te.append((7, 95, 1.0, t))  # an item that only exists in test
    tr = np.array(tr, dtype=np.float64)
    te = np.array(te, dtype=np.float64)
    tr = tr[rng.permutation(len(tr))]  # file order is NOT grouped by user
    os.makedirs(DATA_DIR, exist_ok=True)
    for name, arr in (("train", tr), ("test", te)):
        with open(os.path.join(DATA_DIR, "tiny." + name), "w") as f:
            for u, i, r, ts in arr:
                f.write(f"{int(u)}\t{int(i)}\t{r:.1f}\t{int(ts)}\n")
    np.savez_compressed(os.path.join(HERE, "tiny_dataset.npz"),
                        train=tr[:, [0, 1, 3]].astype(np.int64), test=te[:, [0, 1, 3]].astype(np.int64),
                        num_users=U, num_items=I)
    print("dataset:", len(tr), "train /", len(te), "test")


def _install():
    sys.path.insert(0, HERE)
    import ref_harness
    os.makedirs(SCRATCH, exist_ok=True)
    os.chdir(SCRATCH)  # the reference writes log/ and _data_cache/ relative to cwd / data_dir
    return ref_harness.install()


def _dict_to_csr(d, n):
    rowptr = np.zeros(n + 1, np.int64)
    for u, items in d.items():
        rowptr[u + 1] = len(items)
    rowptr = np.cumsum(rowptr)
    items = np.concatenate([np.asarray(d[u], np.int32) for u in sorted(d)]) if d else np.zeros(0, np.int32)
    return rowptr, items


# ------------------------------------------------------------------------------------------------
def make_sampler():
    _install()
    from skrec.utils.py import randint_choice, batch_randint_choice
    from skrec.io import RSDataset, PairwiseIterator, PointwiseIterator
    out = {}
    # --- known-answer draws, in this order, from a fresh process -------------------------------
    out["ka1"] = randint_choice(1682, size=10, exclusion=[1, 2, 3])
    out["ka2"] = np.int32(randint_choice(1682, size=1, exclusion=[5]))
    out["ka3"] = randint_choice(50, size=20, replace=False, exclusion=[0, 1, 2, 3])
    p = (np.arange(30, dtype=np.float32) % 7 + 1.0)
    out["ka4_p"] = p
    out["ka4"] = randint_choice(30, size=15, p=p)
    out["ka5"] = np.concatenate(batch_randint_choice(40, [3, 5, 2], exclusion=[[1, 2], [3], [4, 5, 6]],
                                                      thread_num=1))
    out["ka6"] = randint_choice(7, size=40)  # no exclusion, small range (Lemire rejections occur)
    # --- iterators over the tiny dataset ----------------------------------------------------------
    ds = RSDataset(DATA_DIR, "\t", "UIRT")
    train = ds.train_data
    ud = train.to_user_dict()
    rowptr, items = _dict_to_csr(ud, ds.num_users)
    out["train_rowptr"], out["train_items_fileorder"] = rowptr, items

    def run(it):
        cols = None
        lens = []
        for batch in it:
            if cols is None:
                cols = [[] for _ in batch]
            for c, b in zip(cols, batch):
                c.append(np.asarray(b))
            lens.append(len(batch[0]))
        return [np.concatenate(c, axis=0) for c in cols], np.array(lens)

    it = PairwiseIterator(train, num_neg=1, batch_size=128, shuffle=False)
    out["pw_len"] = len(it)
    (u, i, j), lens = run(it)
    out["pw_e1_users"], out["pw_e1_pos"], out["pw_e1_neg"], out["pw_e1_lens"] = u, i, j, lens
    (u, i, j), lens = run(it)
    out["pw_e2_neg"] = j
    it3 = PairwiseIterator(train, num_neg=3, batch_size=100, shuffle=False, drop_last=True)
    out["pw3_len"] = len(it3)
    (u, i, j), lens = run(it3)
    out["pw3_users"], out["pw3_pos"], out["pw3_neg"], out["pw3_lens"] = u, i, j, lens
    pt = PointwiseIterator(train, num_neg=2, batch_size=100, shuffle=False)
    out["pt_len"] = len(pt)
    (u, i, l), lens = run(pt)
    out["pt_users"], out["pt_items"], out["pt_labels"], out["pt_lens"] = u, i, l, lens
    np.random.seed(7)
    its = PairwiseIterator(train, num_neg=1, batch_size=128, shuffle=True)
    (u, i, j), lens = run(its)
    out["pws_users"], out["pws_pos"], out["pws_neg"], out["pws_lens"] = u, i, j, lens
    np.savez_compressed(os.path.join(HERE, "golden_sampler.npz"), **out)
    print("sampler: ok", {k: np.shape(v) for k, v in out.items() if k.startswith("pw")})


# ------------------------------------------------------------------------------------------------
def make_eval():
    _install()
    from skrec.utils.py.cython import eval_score_matrix
    from skrec.utils.py import RankingEvaluator
    from skrec.io import RSDataset
    rng = np.random.default_rng(99)
    out = {}
    cases = [  # (B, I, K, max truth)
        (4, 50, 10, 3), (3, 17, 10, 5), (2, 12, 12, 2), (5, 200, 1, 1), (3, 300, 50, 70),
        (6, 64, 5, 0), (1, 1000, 20, 8), (8, 33, 16, 33)]
    out["n_cases"] = len(cases)
    for c, (B, I, K, mt) in enumerate(cases):
        sc = rng.permutation(B * I).reshape(B, I).astype(np.float32) / np.float32(7.0) - np.float32(3.0)
        if c % 2 == 1:  # masked (train) items -> -inf, always fewer than I-K per row
            for b in range(B):
                nm = int(rng.integers(0, max(1, (I - K) // 2)))
                sc[b, rng.choice(I, nm, replace=False)] = -np.inf
        tests = [rng.choice(I, int(rng.integers(0 if mt == 0 else 1, mt + 1)), replace=False).astype(np.int32)
                 if mt > 0 else np.zeros(0, np.int32) for _ in range(B)]
        mids = [1, 2, 3, 4, 5] if c % 3 else [4, 2]
        rows = eval_score_matrix(sc.copy(), tests, mids, K, 2)
        out[f"c{c}_scores"], out[f"c{c}_K"], out[f"c{c}_mids"], out[f"c{c}_rows"] = sc, K, np.int32(mids), rows
        out[f"c{c}_test_rowptr"] = np.cumsum([0] + [len(t) for t in tests]).astype(np.int64)
        out[f"c{c}_test_items"] = np.concatenate(tests) if sum(map(len, tests)) else np.zeros(0, np.int32)
    # --- end-to-end RankingEvaluator over the tiny dataset with a fixed score table --------------
    ds = RSDataset(DATA_DIR, "\t", "UIRT")
    trd, ted = ds.train_data.to_user_dict(), ds.test_data.to_user_dict()
    table = (rng.permutation(ds.num_users * ds.num_items).reshape(ds.num_users, ds.num_items)
             .astype(np.float32) / np.float32(11.0))

    class Fixed:
        def predict(self, users):
            return table[np.asarray(users)].copy()
    out["e2e_table"] = table
    for tag, metric, top_k, bs in (("a", None, (5, 10, 20), 16), ("b", ["Recall", "NDCG"], 7, 64),
                                    ("c", "MRR", [3], 5)):
        ev = RankingEvaluator(trd, ted, metric=metric, top_k=top_k, batch_size=bs, num_thread=2)
        rep = ev.evaluate(Fixed())
        out[f"e2e_{tag}_names"] = np.array(list(rep.metrics()))
        out[f"e2e_{tag}_values"] = np.array(list(rep.values()), np.float32)
        sub = [3, 63, 5, 10, 11, 7]  # includes a train-only user (5, dropped) and a test-only user (63)
        rep2 = ev.evaluate(Fixed(), test_users=sub)
        out[f"e2e_{tag}_sub_values"] = np.array(list(rep2.values()), np.float32)
    out["e2e_sub_users"] = np.int32(sub)
    np.savez_compressed(os.path.join(HERE, "golden_eval.npz"), **out)
    print("eval: ok")


# ------------------------------------------------------------------------------------------------
def _seed_all():
    import random
    import torch
    np.random.seed(SEED); random.seed(SEED); torch.manual_seed(SEED)  # run_skrec.py:8-29


def _run_config(**kw):
    from skrec import RunConfig
    base = dict(recommender="x", data_dir=DATA_DIR, file_column="UIRT", sep="\t", hyperopt=False, gpu_id=0,
                metric=("Precision", "Recall", "MAP", "NDCG", "MRR"), top_k=(5, 10, 20),
                test_batch_size=16, test_thread=2, seed=SEED)
    base.update(kw)
    return RunConfig(**base)


def _record_reports(model):
    reports = []
    orig = model.evaluate

    def wrapped(test_users=None):
        r = orig(test_users)
        reports.append(np.array(list(r.values()), np.float32))
        return r
    model.evaluate = wrapped
    return reports


def make_bprmf():
    _install()
    import torch
    torch.set_num_threads(1)
    import skrec.recommender.BPRMF as M
    _seed_all()
    model = M.BPRMF(_run_config(recommender="BPRMF"), dict(lr=1e-3, reg=1e-3, n_dim=64, batch_size=256, epochs=3))
    out = {"U0": model.mf.user_embeddings.weight.detach().numpy().copy(),
           "V0": model.mf.item_embeddings.weight.detach().numpy().copy(),
           "b0": model.mf.item_biases.weight.detach().numpy().copy()}
    bpr, l2 = [], []
    ob, ol = M.bpr_loss, M.l2_loss

    def rb(a, b):
        r = ob(a, b); bpr.append(float(r.sum())); return r

    def rl(*w):
        r = ol(*w); l2.append(float(r)); return r
    M.bpr_loss, M.l2_loss = rb, rl
    reports = _record_reports(model)
    best = model.fit()
    out.update(bpr_sum=np.float32(bpr), l2=np.float32(l2), reports=np.stack(reports),
               names=np.array(model.evaluator.metrics_list), best=np.array(list(best.values()), np.float32),
               U1=model.mf.user_embeddings.weight.detach().numpy(), V1=model.mf.item_embeddings.weight.detach().numpy(),
               b1=model.mf.item_biases.weight.detach().numpy(),
               pred_users=np.int32([0, 3, 9, 63]), pred=model.predict([0, 3, 9, 63]))
    np.savez_compressed(os.path.join(HERE, "golden_bprmf.npz"), **out)
    print("bprmf: steps", len(bpr), "first/last bpr", bpr[0], bpr[-1], "NDCG@10", dict(best.items())["NDCG@10"])


def make_lightgcn():
    _install()
    import torch
    torch.set_num_threads(1)
    import skrec.recommender.LightGCN as M
    _seed_all()
    model = M.LightGCN(_run_config(recommender="LightGCN"),
                       dict(lr=1e-3, reg=1e-3, embed_size=64, n_layers=3, adj_type="pre", batch_size=256, epochs=2))
    adj = model.lightgcn.norm_adj.coalesce()
    out = {"adj_idx": adj.indices().numpy().copy(), "adj_val": adj.values().numpy().copy(),
           "U0": model.lightgcn.user_embeddings.weight.detach().numpy().copy(),
           "V0": model.lightgcn.item_embeddings.weight.detach().numpy().copy()}
    bpr, l2 = [], []
    ob, ol = M.bpr_loss, M.l2_loss

    def rb(a, b):
        r = ob(a, b); bpr.append(float(r.mean())); return r

    def rl(*w):
        r = ol(*w); l2.append(float(r)); return r
    M.bpr_loss, M.l2_loss = rb, rl
    reports = _record_reports(model)
    best = model.fit()
    model.lightgcn.eval()
    out.update(bpr_mean=np.float32(bpr), l2=np.float32(l2), reports=np.stack(reports),
               names=np.array(model.evaluator.metrics_list), best=np.array(list(best.values()), np.float32),
               U1=model.lightgcn.user_embeddings.weight.detach().numpy(),
               V1=model.lightgcn.item_embeddings.weight.detach().numpy(),
               Uf=model.lightgcn._user_embeddings_final.detach().numpy(),
               Vf=model.lightgcn._item_embeddings_final.detach().numpy(),
               pred_users=np.int32([0, 3, 9, 63]), pred=model.predict([0, 3, 9, 63]))
    for t in ("plain", "norm", "gcmc"):  # the other adjacency flavours (LightGCN.py:150-164)
        a = model._create_adj_mat(t).tocoo()
        order = np.lexsort((a.col, a.row))
        out[f"adj_{t}_idx"] = np.stack([a.row[order], a.col[order]]).astype(np.int64)
        out[f"adj_{t}_val"] = a.data[order].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "golden_lightgcn.npz"), **out)
    print("lightgcn: steps", len(bpr), "first/last", bpr[0], bpr[-1])


def make_layergcn():
    _install()
    import torch
    torch.set_num_threads(1)
    import skrec.recommender.LayerGCN as M
    _seed_all()
    model = M.LayerGCN(_run_config(recommender="LayerGCN"),
                       dict(lr=1e-3, reg=1e-2, embed_dim=64, n_layers=4, dropout=0.0, batch_size=256, epochs=2))
    adj = model.model.norm_adj_matrix.coalesce()
    out = {"adj_idx": adj.indices().numpy().copy(), "adj_val": adj.values().numpy().copy(),
           "U0": model.model.user_embeddings.detach().numpy().copy(),
           "V0": model.model.item_embeddings.detach().numpy().copy()}
    losses = []
    oc = model.model.calculate_loss

    def rc(u, i, j):
        r = oc(u, i, j); losses.append(float(r)); return r
    model.model.calculate_loss = rc
    reports = _record_reports(model)
    best = model.fit()
    with torch.no_grad():
        model.model.forward_adj = model.model.norm_adj_matrix
        uf, vf = model.model.forward()
    out.update(loss=np.float32(losses), reports=np.stack(reports), names=np.array(model.evaluator.metrics_list),
               best=np.array(list(best.values()), np.float32),
               U1=model.model.user_embeddings.detach().numpy(), V1=model.model.item_embeddings.detach().numpy(),
               Uf=uf.numpy(), Vf=vf.numpy(), pred_users=np.int32([0, 3, 9, 63]), pred=model.predict([0, 3, 9, 63]))
    np.savez_compressed(os.path.join(HERE, "golden_layergcn.npz"), **out)
    print("layergcn: steps", len(losses), "first/last", losses[0], losses[-1])


def make_layergcn_dropout():
    """dropout > 0 (LayerGCN.py:133-152): three epochs = degree-weighted pruning (torch.multinomial on the CPU
    generator), uniform pruning (Python's random.sample), degree-weighted again; the pruned + re-normalised
    adjacency of every epoch, the loss trajectory, the reports (evaluation runs on the FULL graph) and the final tables"""
    _install()
    import torch
    torch.set_num_threads(1)
    import skrec.recommender.LayerGCN as M
    _seed_all()
    model = M.LayerGCN(_run_config(recommender="LayerGCN"),
                       dict(lr=1e-3, reg=1e-2, embed_dim=64, n_layers=3, dropout=0.2, batch_size=256, epochs=3))
    out = {"U0": model.model.user_embeddings.detach().numpy().copy(),
           "V0": model.model.item_embeddings.detach().numpy().copy(),
           "edge_values": model.model.edge_values.numpy().copy(),
           "edge_indices": model.model.edge_indices.numpy().copy()}
    losses, masked = [], []
    oc, op = model.model.calculate_loss, model.model.pre_epoch_processing

    def rc(u, i, j):
        r = oc(u, i, j); losses.append(float(r)); return r

    def rp():
        op()
        a = model.model.masked_adj.coalesce()
        masked.append((a.indices().numpy().copy(), a.values().numpy().copy()))
    model.model.calculate_loss, model.model.pre_epoch_processing = rc, rp
    reports = _record_reports(model)
    best = model.fit()
    for e, (idx, val) in enumerate(masked):
        out[f"masked{e}_idx"], out[f"masked{e}_val"] = idx, val
    out.update(loss=np.float32(losses), reports=np.stack(reports), names=np.array(model.evaluator.metrics_list),
               best=np.array(list(best.values()), np.float32), n_epochs=np.int32(len(masked)),
               U1=model.model.user_embeddings.detach().numpy(), V1=model.model.item_embeddings.detach().numpy())
    np.savez_compressed(os.path.join(HERE, "golden_layergcn_dropout.npz"), **out)
    print("layergcn dropout: epochs", len(masked), "steps", len(losses), "kept", [len(v) // 2 for _, v in masked],
          "first/last", losses[0], losses[-1])


def make_iterators():
    """the sequential and knowledge-graph iterators, shuffle=False, one process = one stream;
    tests/test_gpu_iterators.py replays the same constructions in the same order."""
    _install()
    import pandas as pd
    from skrec.io import RSDataset, SequentialPairwiseIterator, SequentialPointwiseIterator, KGPairwiseIterator
    from skrec.io.dataset import KnowledgeGraph
    ds = RSDataset(DATA_DIR, "\t", "UIRT")
    train = ds.train_data
    out = {}

    def run(it):
        cols = None
        for batch in it:
            if cols is None:
                cols = [[] for _ in batch]
            for c, b in zip(cols, batch):
                c.append(np.asarray(b))
        return [np.concatenate(c, axis=0) for c in cols]

    def record(tag, it, epochs=1):
        out[tag + "_len"] = len(it)
        for e in range(epochs):
            for k, col in enumerate(run(it)):
                out[f"{tag}_e{e}_c{k}"] = col

    record("spw", SequentialPairwiseIterator(train, num_previous=3, num_next=1, pad=None, batch_size=128, shuffle=False), 2)
    record("spwpad", SequentialPairwiseIterator(train, num_previous=4, num_next=2, pad=ds.num_items, batch_size=100,
                                                shuffle=False, drop_last=True))
    record("spw11", SequentialPairwiseIterator(train, num_previous=1, num_next=1, pad=None, batch_size=256, shuffle=False))
    record("spt", SequentialPointwiseIterator(train, num_previous=2, num_next=1, num_neg=2, pad=None, batch_size=128,
                                              shuffle=False))
    record("sptpad", SequentialPointwiseIterator(train, num_previous=3, num_next=2, num_neg=2, pad=ds.num_items,
                                                 batch_size=128, shuffle=False))
    rng = np.random.default_rng(5)
    n_ent, n_rel, n_tri = 60, 4, 400
    tri = np.unique(np.stack([rng.integers(0, 40, n_tri), rng.integers(0, n_rel, n_tri), rng.integers(0, n_ent, n_tri)], 1),
                    axis=0)
    tri = tri[rng.permutation(len(tri))]                     # heads in arbitrary file order, duplicate tails per head
    out["kg_triplets"] = tri.astype(np.int32)
    kg = KnowledgeGraph(pd.DataFrame(tri, columns=["head", "relation", "tail"]), num_entities=n_ent, num_relations=n_rel)
    record("kg", KGPairwiseIterator(kg, num_neg=1, batch_size=64, shuffle=False))
    record("kg3", KGPairwiseIterator(kg, num_neg=3, batch_size=64, shuffle=False))
    np.savez_compressed(os.path.join(HERE, "golden_iterators.npz"), **out)
    print("iterators: ok", {k: np.shape(v) for k, v in out.items() if k.endswith("_c0") or k.endswith("_len")})


SECTIONS = {"dataset": make_dataset, "sampler": make_sampler, "eval": make_eval, "bprmf": make_bprmf,
            "lightgcn": make_lightgcn, "layergcn": make_layergcn, "layergcn_dropout": make_layergcn_dropout,
            "iterators": make_iterators}

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "all":
        import shutil
        shutil.rmtree(SCRATCH, ignore_errors=True)  # stale _LightGCN_data/pre_adj.npz would be reused
        for s in SECTIONS:
            subprocess.run([sys.executable, os.path.abspath(__file__), s], check=True)
    else:
        SECTIONS[what]()
