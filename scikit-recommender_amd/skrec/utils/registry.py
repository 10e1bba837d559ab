"""Model discovery by module name (reference: utils/registry.py)."""
from collections import OrderedDict
from importlib import import_module
from importlib.util import find_spec

__all__ = ["ModelRegistry"]


class ModelRegistry:
    """``load_skrec_model("BPRMF")`` imports ``skrec.recommender.BPRMF`` and expects the classes
    ``BPRMF`` and ``BPRMFConfig`` in it (registry.py:17-36)."""

    def __init__(self):
        self.models = OrderedDict()
        self.configs = OrderedDict()

    def register_model(self, model_name, model_class, config_class):
        self.models[model_name] = model_class
        self.configs[model_name] = config_class

    def load_skrec_model(self, model_name: str, spec_path="skrec.recommender") -> bool:
        module_path = f"{spec_path}.{model_name}"
        try:
            found = find_spec(module_path) is not None
        except ModuleNotFoundError:
            found = False
        if not found:
            print(f"Module '{module_path}' is not found.")
            return False
        module = import_module(module_path)
        model_class = getattr(module, model_name, None)
        config_class = getattr(module, f"{model_name}Config", None)
        if model_class is None or config_class is None:
            print(f"Import '{model_name}' or '{model_name}Config' failed from {module.__file__}!")
            return False
        self.register_model(model_name, model_class, config_class)
        return True

    def get_model(self, model_name: str):
        return self.models.get(model_name), self.configs.get(model_name)

    def list_models(self):
        return list(self.models)
