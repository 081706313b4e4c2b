"""Drop-in import surface of the reference package: `from speech_decoding.models import BrainEncoder,
Classifier` and `from speech_decoding.utils.loss import *` (train.py:22,24) resolve to the MI355X build."""
