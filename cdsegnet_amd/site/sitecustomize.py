"""Put this directory (and the repo root) on PYTHONPATH to run the reference's tools unchanged on the MI355X
models: every interpreter of the run re-registers "DefaultSegmentorV2" / "PT-v3m1" right after it imports
``pointcept.models`` (see cdsegnet_amd/pointcept_plugin.py)."""
try:
    from cdsegnet_amd import pointcept_plugin as _plug
    _plug.install_import_hook()
except Exception as _e:  # never break an unrelated interpreter start-up
    import sys as _sys
    _sys.stderr.write(f"cdsegnet_amd sitecustomize: plugin not installed ({_e!r})\n")
