"""Imported automatically by Python's `site` when <repo>/dropin is on PYTHONPATH: installs the import hook of mvs_dropin
(MVS_DROPIN=0 disables it)."""
import os

if os.environ.get("MVS_DROPIN", "1") != "0":
    import mvs_dropin
    mvs_dropin.install()
