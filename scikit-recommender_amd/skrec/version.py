# API-compatible with scikit-recommender v0.1.1 (the surveyed reference revision)
__version__ = "0.1.1+mi355x"
