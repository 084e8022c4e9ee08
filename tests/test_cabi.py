"""The C-ABI library loads and exports every symbol include/slamhip.h declares;
the ctypes table binds exactly those.  No compute calls (CPU-only box)."""
import ctypes
import os
import re

from slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'slamhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(slamhip_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported():
    names = declared_symbols()
    assert 'slamhip_chi2_batch' in names and 'slamhip_create' in names
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in slamhip.h but not exported'


def test_ctypes_table_matches_header():
    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_desc_layout_matches_header():
    """Field order of the ctypes mirror == field order of struct slamhip_desc."""
    text = open(os.path.join(ROOT, 'include', 'slamhip.h')).read()
    body = re.search(r'typedef struct slamhip_desc \{(.*?)\} slamhip_desc;', text, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith('const') else decl.split('*', 1)[1]
        fields += [n.strip().lstrip('*') for n in names.split(',')]
    assert fields == [f[0] for f in _lib.SlamhipDesc._fields_]


def test_library_reports_itself():
    lib = _lib.load()
    assert lib.slamhip_version().decode().startswith('slamhip')
    assert lib.slamhip_kernel_count() >= 5
    assert lib.slamhip_device_count() >= 0
    # bad arguments are refused without touching a GPU
    assert lib.slamhip_create(None, None) < 0
