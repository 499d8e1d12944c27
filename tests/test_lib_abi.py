"""The C-ABI library loads on a machine without a GPU and exports every symbol include/epnn.h declares;
no compute is called. CPU only."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "epnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epnn_[a-z0-9_]+)\s*\(", text)))


def test_build_produces_library():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(g.LIB)


def test_every_declared_symbol_is_exported_and_bound():
    from epnn_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/epnn.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in epnn_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.epnn_version() >= 1


def test_no_silent_cpu_fallback():
    """Without a GPU the product path raises; it never computes on the host."""
    from epnn_amd import _lib
    from epnn_amd.engine import Engine
    if _lib.load().epnn_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.EpnnError, match="no HIP device"):
        Engine()
    src = open(os.path.join(ROOT, "epnn_amd", "charge_gn.py")).read() + open(os.path.join(ROOT, "epnn_amd", "engine.py")).read()
    assert "oracle" not in src


def test_every_option_is_documented_in_the_header():
    """epnn_set_option's names all appear in include/epnn.h (the options a caller may touch: about ten) or in include/epnn_dev.h (the
    developer switches), and in exactly one of the two."""
    import glob
    import re
    src = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "epnn_amd", "csrc", "*.hip"))))
    hdr = open(os.path.join(ROOT, "include", "epnn.h")).read()
    dev = open(os.path.join(ROOT, "include", "epnn_dev.h")).read()
    names = re.findall(r'!strcmp\(name, "([a-z0-9_]+)"\)', src)
    assert len(names) >= 10
    missing = [n for n in names if (f'"{n}"' in hdr) == (f'"{n}"' in dev)]
    assert not missing, missing
    public = [n for n in names if f'"{n}"' in hdr]
    assert len(public) <= 12, public
    assert not re.findall(r"\bepnn_[a-z0-9_]+\s*\(", re.sub(r"/\*.*?\*/", "", dev, flags=re.S)), "epnn_dev.h declares no symbols"
