"""install_as_audiossl(): the reference's own import statements resolve to the HIP-backed modules, every other
``audiossl.*`` import is left to the installed distribution (here: a dummy package tree), and uninstall() undoes it.
Runs in subprocesses so that sys.modules of the test session stays untouched.  No GPU, no libatst_hip.so calls."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree(tmp, files):
    for rel, body in files.items():
        p = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(body))


def _run(code, extra_path=None):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(filter(None, [extra_path, ROOT])))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_aliases_inside_an_installed_distribution(tmp_path):
    # a stand-in for an installed reference: real siblings (datasets), decoys for the modules that must be replaced,
    # a parent __init__ that imports its own submodule (as audiossl/models/atst/__init__.py does), and a downstream-style
    # consumer with the harness's import lines (downstream/model.py:4,9 ; downstream/train_freeze.py:10)
    _tree(str(tmp_path), {
        "audiossl/__init__.py": "TOP = 'dummy-distribution'\n",
        "audiossl/datasets/__init__.py": "REAL = True\n",
        "audiossl/methods/atst/__init__.py": "",
        "audiossl/methods/atst/model.py": "class ATSTLightningModule:\n    decoy = True\n",
        "audiossl/methods/atst/downstream/__init__.py": "",
        "audiossl/methods/atst/downstream/consumer.py": """
            from audiossl.methods.atst.model import ATSTLightningModule
            from audiossl.models.atst import audio_transformer
            from audiossl.utils.common import cosine_scheduler_epoch, get_params_groups
            from audiossl import datasets
            def walk_names():
                return (ATSTLightningModule.load_from_checkpoint, audio_transformer.AST.get_intermediate_layers_chunks,
                        audio_transformer.AST_small, audio_transformer.AST_base)
        """,
        "audiossl/models/__init__.py": "",
        "audiossl/models/atst/__init__.py": "from .atst import ATST\n__all__ = ['ATST']\n",
        "audiossl/models/atst/atst.py": "class ATST:\n    decoy = True\n",
        "audiossl/utils/__init__.py": "",
        "audiossl/utils/common.py": "decoy = True\n",
    })
    out = _run("""
        import sys
        import audiossl_amd
        served = audiossl_amd.install_as_audiossl()
        assert set(served) >= {"audiossl.methods.atst.model", "audiossl.methods.atstframe.model", "audiossl.utils.common"}
        from audiossl.methods.atst.model import ATSTLightningModule
        from audiossl.methods.atstframe.model import FrameATSTLightningModule
        import audiossl_amd.methods.atst.model as mine
        assert ATSTLightningModule is mine.ATSTLightningModule and not hasattr(ATSTLightningModule, "decoy")
        assert FrameATSTLightningModule.__module__ == "audiossl_amd.methods.atstframe.model"
        import audiossl, audiossl.datasets
        assert audiossl.TOP == "dummy-distribution" and audiossl.datasets.REAL          # the rest of the distribution is untouched
        from audiossl.models.atst import ATST                                             # parent __init__ did `from .atst import ATST`
        from audiossl_amd.models.atst.atst import ATST as MineATST
        assert ATST is MineATST
        import audiossl.utils.common as uc
        assert not hasattr(uc, "decoy") and callable(uc.cosine_scheduler_epoch) and callable(uc.concat_all_gather)
        assert len(uc.cosine_scheduler_epoch(1.0, 0.0, 3, 10, warmup_epochs=1)) == 30
        from audiossl.methods.atst.downstream import consumer                            # a module of the distribution importing the aliased names
        assert all(callable(f) for f in consumer.walk_names())
        from audiossl.methods.atst.transform import ATSTTrainTransform
        assert ATSTTrainTransform.__module__ == "audiossl_amd.methods.atst.transform"
        audiossl_amd.compat.uninstall()
        for k in list(sys.modules):
            if k.startswith("audiossl.") or k == "audiossl":
                del sys.modules[k]
        from audiossl.methods.atst.model import ATSTLightningModule as Decoy
        assert Decoy.decoy                                                                # after uninstall the distribution's own module is back
        print("ok")
    """, str(tmp_path))
    assert "ok" in out


def test_aliases_without_any_distribution():
    out = _run("""
        import sys
        assert not any(k == "audiossl" or k.startswith("audiossl.") for k in sys.modules)
        import audiossl_amd
        audiossl_amd.install_as_audiossl()
        audiossl_amd.install_as_audiossl()                                               # idempotent
        from audiossl.methods.atst.model import ATSTLightningModule
        from audiossl.methods.atstframe.embedding import load_model, get_scene_embedding, get_timestamp_embedding
        from audiossl.models.atst.audio_transformer import AST, AST_small, AST_base
        assert ATSTLightningModule.__module__.startswith("audiossl_amd.")
        try:
            import audiossl.datasets                                                     # not aliased, not installed: an honest ImportError
            raise SystemExit("audiossl.datasets should not exist here")
        except ImportError:
            pass
        print("ok")
    """)
    assert "ok" in out


def test_genuine_import_error_in_the_distribution_is_not_hidden(tmp_path):
    # an installed audiossl whose methods/atst/__init__ needs a dependency that is missing here: install_as_audiossl() must surface that
    # error, not replace the package by an empty stand-in (which would also make its siblings, e.g. .downstream, unimportable); and
    # uninstall() takes the attributes it set on REAL parent packages away again
    _tree(str(tmp_path), {
        "audiossl/__init__.py": "",
        "audiossl/methods/__init__.py": "",
        "audiossl/methods/atst/__init__.py": "import a_dependency_that_is_not_installed_here\n",
        "audiossl/methods/atst/model.py": "class ATSTLightningModule:\n    decoy = True\n",
    })
    out = _run("""
        import audiossl_amd
        try:
            audiossl_amd.install_as_audiossl()
            raise SystemExit("the distribution's own ImportError was swallowed")
        except ModuleNotFoundError as e:
            assert e.name == "a_dependency_that_is_not_installed_here", e.name
        print("ok")
    """, str(tmp_path))
    assert "ok" in out
    _tree(str(tmp_path), {"audiossl/methods/atst/__init__.py": ""})
    out = _run("""
        import sys, audiossl_amd
        import audiossl.methods.atst as real
        assert not hasattr(real, "transform")
        audiossl_amd.install_as_audiossl()
        assert real.transform.__name__ == "audiossl_amd.methods.atst.transform" and real.model.__name__ == "audiossl_amd.methods.atst.model"
        audiossl_amd.compat.uninstall()
        assert not hasattr(real, "transform") and not hasattr(real, "model")           # nothing of ours left on the real package
        print("ok")
    """, str(tmp_path))
    assert "ok" in out
