from .base import AbstractRecommender
