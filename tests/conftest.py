import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "scikit-recommender_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


GOLDEN = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture()
def tiny_dir(tmp_path, golden):
    """the tiny dataset of the golden fixtures, written in the reference's TSV format"""
    d = golden("tiny_dataset")
    root = tmp_path / "tiny"
    root.mkdir()
    for split in ("train", "test"):
        rows = d[split]
        with open(root / f"tiny.{split}", "w") as f:
            for u, i, t in rows:
                f.write(f"{int(u)}\t{int(i)}\t1.0\t{int(t)}\n")
    return str(root)


@pytest.fixture(params=["f16x2", "bf16x3", "fp32"])
def fused_mode(request, monkeypatch):
    """the arithmetics of skr_eval_fused_topk (read per call from SKR_FUSED_MODE): f16x2 (the default: two fp16 pieces per operand
    behind a guard, the rows it rejects recomputed by bf16x3), bf16x3 (three bf16 pieces, no condition on the operands), fp32"""
    monkeypatch.setenv("SKR_FUSED_MODE", request.param)
    return request.param
