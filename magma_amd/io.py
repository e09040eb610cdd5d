"""Input layer: the reference's data files into column arrays.

The reference reads its data with `read_csv2` (simulation_study.R:496-531): ';' separated,
',' as the decimal mark, padded numeric fields, then casts `ID` / `ID_dataset` to character.
`read_csv2` here returns the same table as a dict of NumPy columns (what `magma_amd.api` and
`engine.Dataset` take as a db), parsed in one vectorised pass instead of row by row.
"""
import io

import numpy as np

ID_COLUMNS = ("ID", "ID_dataset")


def read_csv2(path, id_columns=ID_COLUMNS):
    """Table at `path` -> {column: array}. Columns named in `id_columns` stay strings (stripped:
    `as.character` of the parsed integer), every other column becomes float64."""
    with open(path, "r", encoding="utf-8") as fh:
        header = fh.readline().rstrip("\r\n")
        body = fh.read()
    names = [h.strip().strip('"') for h in header.split(";")]
    if not body.strip():
        return {n: (np.zeros(0, dtype=str) if n in id_columns else np.zeros(0)) for n in names}
    ids = [k for k, n in enumerate(names) if n in id_columns]
    nums = [k for k, n in enumerate(names) if n not in id_columns]
    out = {}
    if nums:
        # decimal comma -> point, then one C-level parse of the numeric columns
        arr = np.loadtxt(io.StringIO(body.replace(",", ".")), delimiter=";", usecols=nums,
                         dtype=np.float64, ndmin=2, comments=None)
        for c, k in enumerate(nums):
            out[names[k]] = np.ascontiguousarray(arr[:, c])
    if ids:
        arr = np.loadtxt(io.StringIO(body), delimiter=";", usecols=ids, dtype=str, ndmin=2,
                         comments=None)
        for c, k in enumerate(ids):
            out[names[k]] = np.char.strip(np.char.strip(arr[:, c]), '"')
    return {n: out[n] for n in names}


def split_by(table, column="ID_dataset"):
    """{value: sub-table} in order of first appearance (`filter(ID_dataset == d)` for every d in
    one grouping pass)."""
    key = np.asarray(table[column])
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(inv.ravel(), kind="stable")
    bounds = np.concatenate([[0], np.cumsum(np.bincount(inv.ravel(), minlength=uniq.size))])
    res = {}
    for u in np.argsort(first, kind="stable"):
        rows = order[bounds[u]:bounds[u + 1]]
        res[uniq[u]] = {c: np.asarray(v)[rows] for c, v in table.items()}
    return res
