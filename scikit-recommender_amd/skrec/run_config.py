"""Run-level settings (reference: skrec/run_config.py:7-43); unknown keys are accepted and ignored,
new optional keys of this implementation ride in through **kwargs."""
from typing import List, Tuple, Union

from .utils.py.config import Config

__all__ = ["RunConfig"]


class RunConfig(Config):
    def __init__(self, recommender="BPRMF", data_dir="dataset/Beauty_loo_u5_i5", file_column="UIRT", sep="\t",
                 hyperopt=False, gpu_id=0, metric=("Precision", "Recall", "MAP", "NDCG", "MRR"),
                 top_k=(10, 20, 30, 40, 50, 100), test_batch_size=64, test_thread=4, seed=2021, **kwargs):
        super().__init__()
        self.recommender: str = recommender
        self.data_dir: str = data_dir
        self.file_column: str = file_column   # UI, UIR, UIT, UIRT
        self.sep: str = sep
        self.hyperopt: bool = hyperopt
        self.gpu_id = gpu_id
        self.metric: Union[None, str, Tuple[str], List[str]] = metric
        self.top_k: Union[int, List[int], Tuple[int]] = top_k
        self.test_batch_size: int = test_batch_size   # a memory knob in the reference; kept for API parity
        self.test_thread: int = test_thread           # unused on the GPU path
        self.seed = seed
        # extensions (not in the reference): "exact" replays the reference MT19937 stream on the GPU,
        # "fast" uses the slot-keyed xoshiro128++ kernel
        self.sampler_mode: str = kwargs.get("sampler_mode", "exact")

    def _validate(self):
        assert isinstance(self.recommender, str) and self.recommender
        assert isinstance(self.data_dir, str) and self.data_dir
        assert isinstance(self.file_column, str) and self.file_column
        assert isinstance(self.sep, str)
        assert isinstance(self.hyperopt, bool)
        assert isinstance(self.test_batch_size, int) and self.test_batch_size > 0
        assert isinstance(self.test_thread, int) and self.test_thread > 0
        assert isinstance(self.seed, int) and self.seed >= 0
        assert self.sampler_mode in ("exact", "fast")
