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


def test_loaded_library_is_the_product_build_of_these_sources():
    """the library the tests and the bench load carries the hash of
    the sources in this tree and is not the phase-stamp (diagnostic) build."""
    from slam_amd import build
    assert _lib.VARIANT == 'product', 'SLAMHIP_VARIANT set in the test environment'
    assert os.path.basename(_lib.LIB_PATH) == 'libslamhip.so'
    lib = _lib.load()
    sha, stamps = _lib.build_info(lib.slamhip_version().decode())
    assert stamps is False
    assert _lib.build_words(lib.slamhip_version().decode())['variant'] == 'product'
    assert sha == build.src_sha('slamhip')
    # no trace of the diagnostic code in the product binary
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'stamp %d' not in blob
    side = build.read_sidecar(_lib.LIB_PATH)
    assert side is not None and side['src_sha'] == sha and side['stamps'] is False
    assert side['variant'] == 'product'


def test_stale_or_stamps_library_is_refused():
    import pytest
    from slam_amd import build
    good = build.src_sha('slamhip')
    _lib.check_build(f'slamhip 0.3 src={good} stamps=0', 'slamhip', 'x.so')
    with pytest.raises(_lib.SlamhipError, match='other sources'):
        _lib.check_build('slamhip 0.3 src=0123456789abcdef stamps=0', 'slamhip', 'x.so')
    with pytest.raises(_lib.SlamhipError, match='other sources'):
        _lib.check_build('slamhip 0.2 (gfx950, fp64 / fp32)', 'slamhip', 'x.so')   # a round-2 binary
    with pytest.raises(_lib.SlamhipError, match='never the product'):
        _lib.check_build(f'slamhip 0.3 src={good} stamps=1 variant=stamps', 'slamhip', 'x.so')
    with pytest.raises(_lib.SlamhipError, match='never the product'):
        _lib.check_build(f'slamhip 0.3 src={good} stamps=1', 'slamhip', 'x.so')
    with pytest.raises(_lib.SlamhipError, match='never the product'):
        _lib.check_build(f'slamhip 0.3 src={good} stamps=0 variant=exp1', 'slamhip', 'x.so')
    with pytest.raises(_lib.SlamhipError, match='never the product'):
        _lib.check_build(f'slamhip 0.3 src={good} stamps=0', 'slamhip', 'x.so', 'exp1')
    _lib.check_build(f'slamhip 0.3 src={good} stamps=0 variant=exp1', 'slamhip', 'x.so', 'exp1')
    _lib.check_build(f'slamhip 0.3 src={good} stamps=1 variant=stamps', 'slamhip', 'x.so', 'stamps')


def test_build_rebuilds_on_hash_not_mtime(tmp_path, monkeypatch):
    """build() trusts the sidecar's hash + flags, never modification times."""
    from slam_amd import build
    calls = []
    monkeypatch.setattr(build.subprocess, 'check_call', lambda cmd: calls.append(cmd))
    out = build.TARGETS['slamhip'][2]
    build._build_one('slamhip', out, [], False, False)
    assert calls == []                              # sidecar matches: nothing to do
    os.utime(out, (0, 0))                           # an ancient file with the right hash is still right
    build._build_one('slamhip', out, [], False, False)
    assert calls == []
    fake = tmp_path / 'libfake.so'
    fake.write_bytes(b'')
    (tmp_path / 'libfake.so.build.json').write_text(
        '{"src_sha": "0000000000000000", "flags": [], "stamps": false}')
    (tmp_path / 'libfake.so.build.json').write_text(
        '{"src_sha": "0000000000000000", "flags": [], "stamps": false, "variant": "product"}')
    build._build_one('slamhip', str(fake), [], False, False)
    assert len(calls) == 1 and any('SLAM_SRC_SHA' in a for a in calls[0])
    # the stamps build goes to its own file
    assert os.path.basename(build.STAMPS_OUT) == 'libslamhip_stamps.so'
