from .registry import ModelRegistry
