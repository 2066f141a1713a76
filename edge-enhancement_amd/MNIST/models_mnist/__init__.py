"""MNIST model zoo (reference: MNIST/models_mnist/__init__.py, Net2.py, Net2_EE.py, Net2_EE_square.py)."""
from eeadv.models import Net_2, Net2_EE, Net2_EE_square  # noqa: F401

__all__ = ["Net_2", "Net2_EE", "Net2_EE_square"]
