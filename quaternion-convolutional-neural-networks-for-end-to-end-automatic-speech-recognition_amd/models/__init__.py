"""Counterparts of the reference's model builders (models/example_model.py,
models/interspeech_model.py): callers of the quaternion hot path, kept as thin torch modules."""
from .example_model import CNN, DNN
from .interspeech_model import getTimitModel2D, TimitQCNN
