"""The C-ABI library loads without a GPU and exports every symbol include/kmerguts_hip.h declares;
without a device the entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "kmerguts_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(kg_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(native):
    lib = native.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(native.EXPORTS) == names
    assert b"gfx950" in lib.kg_version()


def test_record_layouts_match_header(native):
    assert native.HIT_DTYPE.itemsize == 24 and native.CALL_DTYPE.itemsize == 24 and native.OTU_DTYPE.itemsize == 44
    assert ctypes.sizeof(native.KgParams) == 24
    assert ctypes.sizeof(native.KgStats) == 10 * 8 + 4 * 4 + 2 * 4 + 3 * 4 + 7 * 4


def test_kg_stats_binding_matches_the_c_struct(native, tmp_path):
    """Every field of the ctypes KgStats sits where gcc puts it for include/kmerguts_hip.h."""
    import subprocess
    fields = [f for f, _ in native.KgStats._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "kmerguts_hip.h"\nint main(void){\n' +
                   'printf("%zu\\n", sizeof(kg_stats));\n' +
                   "".join('printf("%%zu\\n", offsetof(kg_stats, %s));\n' % f for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(native.KgStats)
    assert out[1:] == [getattr(native.KgStats, f).offset for f in fields]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(native):
    from kmergutsjava_amd import hotpath
    with pytest.raises(native.KmerGutsNativeError) as ei:
        hotpath.SignatureTable.from_bytes(np.zeros(48, dtype=np.uint8))
    assert "no CPU path" in str(ei.value) or ei.value.code in (-4, -3)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under kmergutsjava_amd/ may reference it."""
    pkg = os.path.join(ROOT, "kmergutsjava_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(dirpath, f)
