"""Run the reference's scripts UNCHANGED on the MI355X hot path.

The reference imports its models as ``from models.mvsnet import MVSNet, mvsnet_loss`` (jdacs/train.py:28, jdacs/eval.py:23),
``from .module import *`` (jdacs/models/mvsnet.py:14), ``from models.network import CVPMVSNet, sL1_loss, MSE_loss``
(jdacs-ms/train.py:24, jdacs-ms/test.py) and ``from models.modules import *`` (jdacs-ms/models/network.py:13), with the
script's own directory first on sys.path -- so a PYTHONPATH overlay cannot shadow them.  ``install()`` puts ONE import hook
in front of sys.meta_path that answers exactly those four module names with the drop-in modules of this repository; every
other import (``models.augmentations``, ``models.seg_dff``, ``datasets``, ``losses``, ``utils``, ``config`` ...: out of scope,
SURVEY.md section 2) still resolves to the reference's own files.

    cd Self-Supervised-MVS/jdacs                       # or jdacs-ms
    python <repo>/dropin/run.py train.py --mode train ...            # launcher, or
    PYTHONPATH=<repo>/dropin:<repo> python train.py --mode train ... # dropin/sitecustomize.py installs the hook at start-up
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REDIRECT = {
    "models.mvsnet": "mvs_amd.jdacs.models.mvsnet",            # jdacs/models/mvsnet.py
    "models.module": "mvs_amd.jdacs.models.module",            # jdacs/models/module.py
    "models.network": "mvs_amd.jdacs_ms.models.network",       # jdacs-ms/models/network.py
    "models.modules": "mvs_amd.jdacs_ms.models.modules",       # jdacs-ms/models/modules.py
}


class _Loader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        if _REPO not in sys.path:
            sys.path.append(_REPO)
        import mvs_amd  # noqa: F401  (repo-root alias of the self-supervised-mvs_amd package)
        return importlib.import_module(self.target)   # the SAME module object under both names

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname in REDIRECT:
            return importlib.util.spec_from_loader(fullname, _Loader(REDIRECT[fullname]))
        return None


def install():
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
