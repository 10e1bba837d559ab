"""``HyperOpt`` driver (reference: utils/hyperopt.py).  Without ``--hyperopt True`` it builds the model
and calls ``fit()`` (hyperopt.py:81-83).  The TPE search needs the third-party ``hyperopt`` package,
which is imported lazily so that plain runs work without it."""
import time
from copy import deepcopy

__all__ = ["HyperOpt"]


class HyperOpt(object):
    def __init__(self, run_config, model_class, config_class, fixed_params):
        run_config.hyperopt = run_config.hyperopt and bool(config_class.param_space())
        self._run_config = run_config
        self._model_class = model_class
        self._config_class = config_class
        self._fixed_params = fixed_params
        self._current_model = None
        self._trials = []

    def _fit_once(self, params):
        merged = deepcopy(self._fixed_params)
        merged.update(params)
        self._current_model = self._model_class(self._run_config, merged)
        return self._current_model.fit()

    def run(self):
        if not self._run_config.hyperopt:
            return self._fit_once({})
        try:
            from hyperopt import fmin, tpe, hp, Trials, space_eval
        except ImportError as e:
            raise ImportError("--hyperopt True needs the 'hyperopt' package (not installed here)") from e
        space = {k: hp.choice(k, v) for k, v in self._config_class.param_space().items()}
        n_combos = self._config_class.num_combos()
        key = "NDCG@10"

        def objective(params):
            t0 = time.time()
            report = self._fit_once(params)
            self._trials.append((dict(params), report, time.time() - t0))
            return -float(report[key])
        trials = Trials()
        best = fmin(objective, space, algo=tpe.suggest, max_evals=n_combos, trials=trials)
        best_params = space_eval(space, best)
        print(f"best hyper-parameters: {best_params}")
        return max(self._trials, key=lambda t: float(t[1][key]))[1]
