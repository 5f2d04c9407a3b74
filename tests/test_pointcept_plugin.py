"""The drop-in boundary on the registry side: the MI355X classes go INTO the reference's registry,
so tools/test_*.py (which builds through pointcept.models.builder.MODELS) needs no edit.  A stand-in `pointcept`
package with the reference Registry's interface (register_module(name, force, module) / build(cfg)) is created in a
temp dir; when /root/reference exists (build container only) the reference's real Registry class is used too."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STANDIN_BUILDER = '''
class Registry:  # interface of pointcept/utils/registry.py:Registry that the plugin relies on
    def __init__(self, name):
        self.name, self._module_dict = name, {}
    def get(self, key):
        return self._module_dict.get(key)
    def register_module(self, name=None, force=False, module=None):
        assert isinstance(force, bool)
        def reg(cls):
            for n in ([name] if isinstance(name, str) else (name or [cls.__name__])):
                if not force and n in self._module_dict:
                    raise KeyError(f"{n} is already registered in {self.name}")
                self._module_dict[n] = cls
            return cls
        return reg(module) if module is not None else reg
    def build(self, cfg):
        cfg = dict(cfg)
        return self._module_dict[cfg.pop("type")](**cfg)
MODELS = Registry("models")
def build_model(cfg):
    return MODELS.build(cfg)
'''

STANDIN_INIT = '''
from .builder import MODELS, build_model
@MODELS.register_module("DefaultSegmentorV2")
class DefaultSegmentorV2:  # what the reference registers at import time
    origin = "reference"
@MODELS.register_module("PT-v3m1")
class PointTransformerV3:
    origin = "reference"
'''


def _fake_pointcept(tmp_path):
    pkg = tmp_path / "pointcept" / "models"
    pkg.mkdir(parents=True)
    (tmp_path / "pointcept" / "__init__.py").write_text("")
    (pkg / "builder.py").write_text(STANDIN_BUILDER)
    (pkg / "__init__.py").write_text(STANDIN_INIT)
    return str(tmp_path)


def test_register_into_standin_registry():
    ns = {}
    exec(STANDIN_BUILDER, ns)
    reg = ns["MODELS"]
    reg.register_module("DefaultSegmentorV2", module=type("Ref", (), {}))
    from cdsegnet_amd import configs, pointcept_plugin
    import cdsegnet_amd.models as M
    with pytest.raises(KeyError):
        pointcept_plugin.register_into(reg, force=False)  # the reference's name is taken: force is needed
    pointcept_plugin.register_into(reg)
    assert reg.get("DefaultSegmentorV2") is M.DefaultSegmentorV2 and reg.get("PT-v3m1") is M.PointTransformerV3
    model = reg.build(configs.mini_config())  # cfg.model goes through the REFERENCE-side registry object
    assert isinstance(model, M.DefaultSegmentorV2) and isinstance(model.backbone, M.PointTransformerV3)


def test_site_hook_registers_after_pointcept_models_import(tmp_path):
    """PYTHONPATH=<repo>/cdsegnet_amd/site:<repo> python <unchanged reference tool>: simulated with a stand-in
    pointcept package; the hook must fire after pointcept.models registered its own classes, in a fresh interpreter."""
    fake = _fake_pointcept(tmp_path)
    code = textwrap.dedent('''
        import pointcept.models as pm            # what engines/test.py does first
        import cdsegnet_amd.models as M
        assert pm.MODELS.get("DefaultSegmentorV2") is M.DefaultSegmentorV2, pm.MODELS.get("DefaultSegmentorV2")
        assert pm.MODELS.get("PT-v3m1") is M.PointTransformerV3
        from cdsegnet_amd import configs
        m = pm.build_model(configs.mini_config())
        assert type(m).__module__ == "cdsegnet_amd.models"
        print("PLUGIN_OK")
    ''')
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "cdsegnet_amd", "site"), ROOT, fake])
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "PLUGIN_OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.exists("/root/reference/pointcept/utils/registry.py"),
                    reason="the reference tree only exists in the build container")
def test_register_into_the_reference_registry_class(tmp_path):
    """Same, against the reference's real Registry / build_from_cfg (fresh interpreter: keeps `pointcept` out of
    this process)."""
    code = textwrap.dedent('''
        import sys
        sys.path.insert(0, "/root/reference")
        from pointcept.utils.registry import Registry
        reg = Registry("models")
        @reg.register_module("DefaultSegmentorV2")
        class Ref:
            pass
        from cdsegnet_amd import configs, pointcept_plugin
        import cdsegnet_amd.models as M
        pointcept_plugin.register_into(reg)
        model = reg.build(configs.mini_config())          # the reference's build_from_cfg
        assert isinstance(model, M.DefaultSegmentorV2)
        try:
            reg.build(dict(configs.mini_config(), no_such_kwarg=1))
        except TypeError as e:                             # ctor errors keep the reference's surface
            assert "DefaultSegmentorV2" in str(e)
        else:
            raise AssertionError("bad kwarg accepted")
        print("REF_REGISTRY_OK")
    ''')
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT
    script = tmp_path / "use_ref_registry.py"  # a real file: the reference's Registry infers its scope from the caller's module
    script.write_text(code)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0 and "REF_REGISTRY_OK" in out.stdout, out.stderr[-2000:]
