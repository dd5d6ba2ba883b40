"""Host-side file helpers with the reference's names (utils/file_io.py:8-71): python-file configs, logger, list files.

Config files are plain Python modules defining `cfg` (an EasyDict) and are imported by path, so existing
`train_config.py` / `infer_config.py` files keep working.  They do `from easydict import EasyDict`; when the real
package is not installed a tiny attribute-dict stand-in is registered under that name first.
"""
import importlib
import logging
import os
import sys
import types


class _AttrDict(dict):
    """minimal EasyDict: attribute access, nested dicts converted on assignment"""

    def __init__(self, d=None, **kwargs):
        super(_AttrDict, self).__init__()
        d = dict(d or {}, **kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, dict) and not isinstance(value, _AttrDict):
            value = _AttrDict(value)
        elif isinstance(value, (list, tuple)):
            value = type(value)(_AttrDict(v) if isinstance(v, dict) and not isinstance(v, _AttrDict) else v for v in value)
        super(_AttrDict, self).__setitem__(name, value)

    __setitem__ = __setattr__

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


def ensure_easydict():
    """make `from easydict import EasyDict` importable (uses the real package when present)"""
    try:
        import easydict  # noqa: F401
    except ImportError:
        shim = types.ModuleType('easydict')
        shim.EasyDict = _AttrDict
        sys.modules['easydict'] = shim


def load_config(pyfile):
    """import a python file as a config module and return its `cfg` (reference: file_io.py:8-28)"""
    assert os.path.isfile(pyfile), 'config file does not exist: {}'.format(pyfile)
    ensure_easydict()
    dirname, basename = os.path.dirname(os.path.abspath(pyfile)), os.path.basename(pyfile)
    modulename, _ = os.path.splitext(basename)
    need_reload = modulename in sys.modules
    sys.path.insert(0, dirname)
    try:
        lib = importlib.import_module(modulename)
        if need_reload:
            lib = importlib.reload(lib)
    finally:
        del sys.path[0]
    return lib.cfg


def setup_logger(log_file, name):
    """logger that writes to stdout and to `log_file` (reference: file_io.py:31-58)"""
    os.makedirs(os.path.dirname(os.path.abspath(log_file)), exist_ok=True)
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    logger.handlers = []
    fmt = logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    for handler in (logging.FileHandler(log_file), logging.StreamHandler(sys.stdout)):
        handler.setLevel(logging.INFO)
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    return logger


def readlines(file):
    """non-empty stripped lines of a text file (reference: file_io.py:61-71)"""
    with open(file, 'r') as fp:
        return [ln.strip() for ln in fp if ln.strip()]
