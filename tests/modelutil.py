"""Small model configurations shared by the model-level tests, the smoke test and bench.py."""
import torch


from athena_amd.data.synthetic import DataDescriptions, model_config  # noqa: F401,E402


def oracle_cfg(mc, num_class_plus1, feat_dim):
    c = dict(mc["model_config"])
    c.update(num_class=num_class_plus1, feat_dim=feat_dim, mtl_weight=mc["mtl_weight"])
    return c


TINY = dict(vocab=10, feat=12, mc=model_config(32, 16, 4, 2, 1, 64), B=3, T=37, U=4)
SMALL = dict(vocab=30, feat=20, mc=model_config(64, 32, 2, 2, 2, 128), B=4, T=61, U=6)
