"""The `speech_decoding/` shim must OVERLAY a reference checkout that comes later on PYTHONPATH:
the hot-path modules resolve to this build, everything else (dataset classes, loaders, seeding helpers — what the
reference's train.py:15-25 imports) to the reference tree.  Checked against a stub tree with the reference's shape
(regular package `speech_decoding/`, no `__init__.py` in `dataclass/` and `utils/`)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_overlays_a_reference_checkout(tmp_path):
    ref = tmp_path / "ref" / "speech_decoding"
    (ref / "dataclass").mkdir(parents=True)
    (ref / "utils").mkdir()
    (ref / "__init__.py").write_text("")
    (ref / "models.py").write_text("BrainEncoder = Classifier = 'reference'\n")
    (ref / "dataclass" / "brennan2018.py").write_text("class Brennan2018Dataset: origin = 'reference'\n")
    (ref / "utils" / "reproducibility.py").write_text("def seed_worker(i): return 'reference'\n")
    (ref / "utils" / "get_dataloaders.py").write_text("def get_dataloaders(): return 'reference'\n")
    (ref / "utils" / "loss.py").write_text("CLIPLoss = 'reference'\n")
    code = textwrap.dedent("""
        from speech_decoding.dataclass.brennan2018 import Brennan2018Dataset
        from speech_decoding.models import BrainEncoder, Classifier
        from speech_decoding.utils.get_dataloaders import get_dataloaders
        from speech_decoding.utils.loss import *
        from speech_decoding.utils.reproducibility import seed_worker
        import speech_decoding_amd
        assert Brennan2018Dataset.origin == 'reference' and seed_worker(0) == 'reference' and get_dataloaders() == 'reference'
        assert BrainEncoder is speech_decoding_amd.BrainEncoder and Classifier is speech_decoding_amd.Classifier
        assert CLIPLoss is speech_decoding_amd.CLIPLoss
        print('overlay ok')
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, str(tmp_path / "ref")]), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and "overlay ok" in out.stdout, out.stderr[-2000:]


def test_shim_alone_still_imports():
    code = "from speech_decoding.models import BrainEncoder; from speech_decoding.utils.loss import *; from speech_decoding.utils.layout import ch_locations_2d; print(CLIPLoss.__module__)"
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "speech_decoding_amd.loss" in out.stdout, out.stderr[-2000:]
