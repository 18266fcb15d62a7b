"""`procyon.data.constants`: which description columns of a ProCyon-Instruct text table feed each task
(reference: procyon/data/constants.py: ENTITY_DESCRIPTION_NAMES, QA_SUBSETS, RETRIEVAL_SUBSETS, CAPTION_SUBSETS -- data
tables keyed by subset version and dataset).

The tables are DATA of the training run, not code of this engine, and are not restated here: they are read, on first use, from
`$PROCYON_SUBSETS_JSON` or `$DATA_DIR/procyon_column_subsets.json` -- a JSON with the keys `ENTITY_DESCRIPTION_NAMES`,
`QA_SUBSETS`, `RETRIEVAL_SUBSETS`, `CAPTION_SUBSETS` holding the reference's dictionaries (INTEGRATION.md shows the one-liner
that exports them from a reference checkout).  Without that file every column of the table whose name starts with
"description" (plus "allDescriptions") counts, in table order -- the superset the reference's version-1 subsets draw from."""
import json
import os

_cache = None


def _tables():
    global _cache
    if _cache is None:
        path = os.getenv("PROCYON_SUBSETS_JSON") or (os.path.join(os.getenv("DATA_DIR"), "procyon_column_subsets.json") if os.getenv("DATA_DIR") else None)
        _cache = {}
        if path and os.path.exists(path):
            raw = json.load(open(path))
            for k, v in raw.items():
                _cache[k] = {int(ver) if str(ver).lstrip("-").isdigit() else ver: cols for ver, cols in v.items()} if k.endswith("_SUBSETS") else v
    return _cache


def _default_columns(text_info):
    return [c for c in text_info.columns if str(c).startswith("description") or c == "allDescriptions"]


def entity_description_names(text_type, text_info):
    t = _tables().get("ENTITY_DESCRIPTION_NAMES")
    return t[text_type] if t and text_type in t else _default_columns(text_info)


class _Subsets:
    """`X_SUBSETS[version][dataset]` -> list of columns; falls back to the table's description columns."""

    def __init__(self, key):
        self.key = key

    def __getitem__(self, version):
        outer = self

        class _ByDataset(dict):
            def __missing__(self, dataset):
                raise KeyError(f"{outer.key}[{version!r}][{dataset!r}]: export the reference's column tables (procyon/data/constants.py) "
                               "to $DATA_DIR/procyon_column_subsets.json, see INTEGRATION.md")
        t = _tables().get(self.key)
        if t is not None and version in t:
            return _ByDataset(t[version])
        return _AllDescriptions()


class _AllDescriptions:
    """column_subset stand-in used when no table file exists: resolves against the data frame at hand"""
    frame = None

    def __getitem__(self, dataset):
        return _DeferredColumns()


class _DeferredColumns(list):
    deferred = True


QA_SUBSETS = _Subsets("QA_SUBSETS")
RETRIEVAL_SUBSETS = _Subsets("RETRIEVAL_SUBSETS")
CAPTION_SUBSETS = _Subsets("CAPTION_SUBSETS")
