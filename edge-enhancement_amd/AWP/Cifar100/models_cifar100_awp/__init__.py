from .utils_awp import AdvWeightPerturb, add_into_weights, diff_in_weights  # noqa: F401
