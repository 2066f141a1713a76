"""AWP/Cifar100/models_cifar100_awp/utils_awp.py of the reference is byte-identical to the Tiny-ImageNet one: one implementation."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "Tiny_imagenet"))
from models_tiny_awp.utils_awp import EPS, AdvWeightPerturb, add_into_weights, diff_in_weights  # noqa: E402,F401
