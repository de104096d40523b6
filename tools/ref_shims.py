"""Harness-side shims that let the *unmodified* reference package be imported in the
build container.  Used ONLY by tools/capture_goldens.py to generate the
fixtures under tests/golden/.  Nothing here (and nothing under /root/reference) travels to
the GPU box; tests, smoke() and bench.py never import this module.

Shims (none of them edits a reference file):
  1. `bidict` stand-in  - dict with an `.inverse` view (reference pins bidict==0.18.3, absent here)
  2. synthetic `hwer` package whose __path__ points at /root/reference/hwer so that
     hwer/__init__.py (which eagerly imports dgl through gcn_ncf) is skipped
  3. stub `seaborn`, `more_itertools` (imported by hwer/validation.py)
  4. OneHotEncoder(sparse=...) -> sparse_output (scikit-learn >= 1.4)
"""
import importlib
import itertools
import sys
import types

REF = "/root/reference"


class _Inverse:
    def __init__(self, fwd):
        self._fwd = fwd

    def _table(self):
        f = self._fwd
        if f._inv_len != len(f):
            f._inv = {v: k for k, v in dict.items(f)}
            f._inv_len = len(f)
        return f._inv

    def __getitem__(self, k):
        return self._table()[k]

    def __contains__(self, k):
        return k in self._table()

    def __len__(self):
        return len(self._fwd)


class bidict(dict):
    """Minimal stand-in: construction, update, [], in, keys/items, len, .inverse[...]."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._inv = {}
        self._inv_len = -1

    @property
    def inverse(self):
        return _Inverse(self)


def install():
    if "hwer" in sys.modules and getattr(sys.modules["hwer"], "__graft_ref_shim__", False):
        return sys.modules["hwer"]
    import matplotlib
    matplotlib.use("Agg")

    m = types.ModuleType("bidict")
    m.bidict = bidict
    sys.modules["bidict"] = m

    sns = types.ModuleType("seaborn")
    sys.modules["seaborn"] = sns

    mi = types.ModuleType("more_itertools")
    mi.flatten = itertools.chain.from_iterable

    def chunked(it, n):
        it = iter(it)
        while True:
            c = list(itertools.islice(it, n))
            if not c:
                return
            yield c

    mi.chunked = chunked
    sys.modules["more_itertools"] = mi

    pkg = types.ModuleType("hwer")
    pkg.__path__ = [REF + "/hwer"]
    pkg.__graft_ref_shim__ = True
    sys.modules["hwer"] = pkg

    utils = importlib.import_module("hwer.utils")
    rb = importlib.import_module("hwer.recommendation_base")
    cr = importlib.import_module("hwer.content_recommender")
    embed = importlib.import_module("hwer.embed.BaseEmbed")
    val = importlib.import_module("hwer.validation")

    import sklearn.preprocessing as skp

    def _OHE(sparse=True, **kw):
        kw.setdefault("sparse_output", sparse)
        return skp.OneHotEncoder(**kw)

    cr.OneHotEncoder = _OHE
    pkg.utils, pkg.recommendation_base, pkg.content_recommender, pkg.validation = utils, rb, cr, val
    pkg.embed_base = embed
    return pkg
