"""Command-line driver with the reference's interface (reference: run_skrec.py):

    python run_skrec.py --recommender BPRMF --data_dir dataset/ml-100k --epochs 50 ...

Every ``--key value`` pair is merged into both the run config and the model config, like the
reference (run_skrec.py:58, :74).  ``--sampler_mode fast`` is an extension (see RunConfig).
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from skrec import ModelRegistry, RunConfig, merge_config_with_cmd_args  # noqa: E402
from skrec.utils.hyperopt import HyperOpt  # noqa: E402


def _set_random_seed(seed=2020):
    """numpy / python / torch seeds (run_skrec.py:8-29).  As in the reference this does NOT touch the
    sampler's MT19937(2020) stream."""
    import torch
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main():
    run_dict = {"recommender": "BPRMF", "data_dir": "dataset/ml-100k", "file_column": "UIRT", "sep": "\t",
                "hyperopt": False, "gpu_id": 0, "metric": ("Precision", "Recall", "MAP", "NDCG"),
                "top_k": (10, 20, 30, 40, 50), "test_thread": 4, "test_batch_size": 64, "seed": 2021}
    run_dict = merge_config_with_cmd_args(run_dict)
    run_config = RunConfig(**run_dict)
    name = run_config.recommender
    registry = ModelRegistry()
    registry.load_skrec_model(name)
    if os.path.exists("unarchived_models"):
        registry.load_skrec_model(name, "unarchived_models")
    model_class, config_class = registry.get_model(name)
    if not model_class:
        print(f"Recommender '{name}' is not found.")
        return 1
    model_params = merge_config_with_cmd_args({"lr": 1e-3, "epochs": 500})
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(run_config.gpu_id))
    _set_random_seed(run_config.seed)
    HyperOpt(run_config, model_class, config_class, model_params).run()
    return 0


if __name__ == "__main__":
    sys.exit(main())
