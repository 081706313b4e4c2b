"""Drop-in import surface of the reference package: `from speech_decoding.models import BrainEncoder,
Classifier` and `from speech_decoding.utils.loss import *` (train.py:22,24) resolve to the MI355X build.

This directory OVERLAYS the reference's own `speech_decoding` package instead of shadowing it: with this repo
ahead of the reference checkout on PYTHONPATH, `__path__` below is extended by every other `speech_decoding/`
directory on sys.path, so the modules this build does not replace (`speech_decoding.dataclass.*`,
`speech_decoding.utils.get_dataloaders`, `.reproducibility`, `.preproc_utils`, ... — train.py:15-25) still resolve
to the reference's files, while `models`, `utils.loss` and `utils.layout` are found here first.
(`utils/` deliberately has no `__init__.py`, like the reference's: it is a namespace package over both trees.)"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
