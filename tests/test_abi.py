"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/seg3d_hip.h declares, and the ctypes table matches the header (no compute calls without a GPU)."""
import ctypes
import os
import re

from conftest import REPO, PKG


def _declared_symbols():
    text = open(os.path.join(REPO, 'include', 'seg3d_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(seg3d_[a-zA-Z0-9_]+)\s*\(', text)))


def test_library_built_and_exports_every_declared_symbol():
    import __graft_entry__  # noqa: F401  (repo root is on sys.path)
    from segmentation3d import _engine
    if not os.path.isfile(_engine.LIB_PATH):
        __graft_entry__.build()
    lib = ctypes.CDLL(_engine.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), 'libseg3d_hip.so does not export {}'.format(n)


def test_ctypes_table_matches_header():
    from segmentation3d import _engine
    assert _engine.symbols() == _declared_symbols()
    lib = _engine.lib()
    assert lib.seg3d_abi_version() == 1
    assert lib.seg3d_target_arch() == b'gfx950'
    # size helpers are pure host arithmetic and must work without a GPU
    assert _engine.query('seg3d_gn_stats_count', 16384 * 3 + 1) == 4
    assert _engine.query('seg3d_packed_mfma_floats', 32, 32, 27) == 27 * 4 * 256
    assert _engine.query('seg3d_conv3d_k3_mfma_wgrad_workspace_floats', 1, 8, 8, 16, 32, 32) % (27 * 1024) == 0


def test_argument_counts_match_header():
    """every prototype in the header has as many parameters as the ctypes argtypes list"""
    from segmentation3d import _engine
    text = open(os.path.join(REPO, 'include', 'seg3d_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for m in re.finditer(r'\b(seg3d_[a-zA-Z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ('', 'void') else len(params.split(','))
        assert n == len(_engine._SIGNATURES[name][1]), name


def test_kernel_sources_are_gfx950_only():
    """no CUDA shims / dual paths in the native sources"""
    csrc = os.path.join(PKG, 'csrc')
    for f in os.listdir(csrc):
        text = open(os.path.join(csrc, f)).read()
        assert '__HIP_PLATFORM_AMD__' not in text and 'cuda_runtime' not in text, f
