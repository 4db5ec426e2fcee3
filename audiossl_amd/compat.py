"""``install_as_audiossl()``: make the reference's own import statements resolve to the HIP-backed classes, without
editing the reference.

The reference's downstream harness and data modules import the pre-training classes by their upstream names
(``from audiossl.methods.atst.model import ATSTLightningModule`` -- audiossl/methods/atst/downstream/train_freeze.py:10,
``from audiossl.methods.atstframe.model import FrameATSTLightningModule`` -- audiossl/methods/atstframe/embedding.py:2,
``from audiossl.models.atst import audio_transformer`` / ``from audiossl.utils.common import ...`` --
audiossl/methods/atst/downstream/model.py:4,9, ``from audiossl.methods.atst.transform import ATSTTrainTransform`` --
audiossl/methods/atst/data.py:4).  After ``audiossl_amd.install_as_audiossl()`` exactly those module names are entries of
``sys.modules`` that point at this package's modules; every OTHER ``audiossl.*`` import (datasets, lightning utilities,
downstream heads, transforms.common, ...) is left to whatever ``audiossl`` distribution is installed.  When none is, empty
stand-in packages are created for the parents so that the aliased names still import.

Nothing here touches the GPU or loads libatst_hip.so; the aliased modules do that when their classes are instantiated.
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Dict, List

#: upstream module name -> module of this package that takes its place
ALIASES: Dict[str, str] = {
    "audiossl.methods.atst.model": "audiossl_amd.methods.atst.model",
    "audiossl.methods.atst.transform": "audiossl_amd.methods.atst.transform",
    "audiossl.methods.atstframe.model": "audiossl_amd.methods.atstframe.model",
    "audiossl.methods.atstframe.transform": "audiossl_amd.methods.atstframe.transform",
    "audiossl.methods.atstframe.embedding": "audiossl_amd.methods.atstframe.embedding",
    "audiossl.models.atst.atst": "audiossl_amd.models.atst.atst",
    "audiossl.models.atst.audio_transformer": "audiossl_amd.models.atst.audio_transformer",
    "audiossl.utils.common": "audiossl_amd.utils.common",
}
_installed: List[str] = []          # sys.modules keys this module created (aliases and stand-in parents)
_saved: Dict[str, types.ModuleType] = {}   # entries that were displaced
_MISSING = object()
_parent_attrs: List[tuple] = []     # (parent module, attribute, previous value or _MISSING): what install set on REAL parent packages


def _ensure_parent(name: str) -> types.ModuleType:
    """The package `name` as the installed ``audiossl`` distribution defines it, or an empty stand-in package."""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)           # aliases are registered first: a parent's __init__ that imports them gets ours
    except ModuleNotFoundError as e:
        # only "this package (or the distribution above it) is not installed" turns into a stand-in; a genuine error inside an installed
        # audiossl -- a missing optional dependency of its __init__, a syntax error -- propagates instead of being hidden behind an empty
        # package that would also make its siblings unimportable
        if e.name is None or not (name == e.name or name.startswith(e.name + ".")):
            raise
        sys.modules.pop(name, None)
        pkg = types.ModuleType(name)
        pkg.__path__ = []                               # a package with nothing of its own to find
        pkg.__package__ = name
        pkg.__doc__ = "stand-in created by audiossl_amd.install_as_audiossl(): the reference distribution is not importable here"
        sys.modules[name] = pkg
        _installed.append(name)
        return pkg


def install_as_audiossl(verbose: bool = False) -> Dict[str, types.ModuleType]:
    """Register the aliases (idempotent).  Returns {upstream module name: module now serving it}."""
    out = {}
    for up, mine in ALIASES.items():                    # 1. the aliased leaves, before any parent package runs its __init__
        mod = importlib.import_module(mine)
        if sys.modules.get(up) is not mod:
            if up in sys.modules:
                _saved[up] = sys.modules[up]
            sys.modules[up] = mod
            _installed.append(up)
        out[up] = mod
    for up, mod in out.items():                         # 2. parents: the real packages where they import, stand-ins elsewhere
        parts = up.split(".")
        for i in range(1, len(parts)):
            parent = _ensure_parent(".".join(parts[:i]))
            child = ".".join(parts[:i + 1])
            if child in sys.modules and getattr(parent, parts[i], _MISSING) is not sys.modules[child]:
                if parent.__name__ not in _installed:   # a real package: remember what uninstall() has to put back
                    _parent_attrs.append((parent, parts[i], getattr(parent, parts[i], _MISSING)))
                setattr(parent, parts[i], sys.modules[child])
    if verbose:
        for up, mod in out.items():
            print(f"{up} -> {mod.__name__}")
    return out


def uninstall() -> None:
    """Undo install_as_audiossl(): drop the aliases and stand-ins, restore displaced modules and the attributes of real parent packages."""
    for parent, attr, old in reversed(_parent_attrs):
        if old is _MISSING:
            if hasattr(parent, attr):
                delattr(parent, attr)
        else:
            setattr(parent, attr, old)
    _parent_attrs.clear()
    for name in reversed(_installed):
        sys.modules.pop(name, None)
    _installed.clear()
    for name, mod in _saved.items():
        sys.modules[name] = mod
    _saved.clear()
