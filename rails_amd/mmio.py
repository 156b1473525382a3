"""MatrixMarket I/O and parameter files for the driver.

The reference's driver reads `A.mtx`, `B.mtx`, `M.mtx` with EpetraExt::MatrixMarketFileToCrsMatrix and writes `V.mtx`,
`T.mtx` with EpetraExt::MultiVectorToMatrixMarketFile (src/main.cpp:66-68,123-126); its solver parameters come from the
"Lyapunov Solver" sublist of a Teuchos XML parameter file (src/main.cpp:57-60,111).  This module reads and writes the same
file formats: `%%MatrixMarket matrix coordinate real|integer|pattern general|symmetric|skew-symmetric` -> CSR (duplicate entries
are summed, as Epetra's insertion does), `%%MatrixMarket matrix array real general` <-> dense column-major arrays, and the
Teuchos `<ParameterList>` XML (or a JSON object with the same names).

The data set of the reference's application test (matlab/DataErik/, read by matlab/test/test_MOC.m:94-121) uses a second, plain
text format: `<name>.info` (n nnz), `<name>.beg` (n + 1 row starts, 1-based), `<name>.jco` (column indices, 1-based), `<name>.co`
(values), and single-column `.co` files for diagonals and vectors; `read_csr` / `read_dense` take those too (a path without
`.mtx` whose `.beg` file exists; a `.co` file).

Host-side only; nothing here touches the GPU.
"""
import io
import json
import os
import xml.etree.ElementTree as ET

import numpy as np


class MatrixMarketError(ValueError):
    pass


def _header(f):
    first = f.readline()
    if not first.lower().startswith("%%matrixmarket"):
        raise MatrixMarketError("not a MatrixMarket file (missing %%MatrixMarket banner)")
    tok = first.split()
    if len(tok) < 5:
        raise MatrixMarketError("short MatrixMarket banner: %r" % first.strip())
    obj, fmt, field, sym = (t.lower() for t in tok[1:5])
    if obj != "matrix" or fmt not in ("coordinate", "array"):
        raise MatrixMarketError("unsupported MatrixMarket object/format: %s %s" % (obj, fmt))
    if field not in ("real", "integer", "pattern", "double"):
        raise MatrixMarketError("unsupported MatrixMarket field: %s (complex matrices are not part of this path)" % field)
    if sym not in ("general", "symmetric", "skew-symmetric"):
        raise MatrixMarketError("unsupported MatrixMarket symmetry: %s" % sym)
    line = f.readline()
    while line and (line.startswith("%") or not line.strip()):
        line = f.readline()
    if not line:
        raise MatrixMarketError("missing size line")
    return fmt, field, sym, [int(x) for x in line.split()]


def _numbers(f, ncols):
    """All remaining numbers of the file as an (n, ncols) float64 array; pandas' C parser when available (27M-line files)."""
    text = f.read()
    try:
        import pandas as pd

        df = pd.read_csv(io.StringIO(text), sep=r"\s+", header=None, comment="%", dtype=np.float64, engine="c", float_precision="round_trip")
        arr = df.to_numpy()
        if arr.ndim == 2 and arr.shape[1] == ncols:
            return np.ascontiguousarray(arr)
        flat = arr[~np.isnan(arr)]
    except Exception:  # noqa: BLE001 - any parser trouble: plain numpy fallback below
        flat = np.array([float(t) for l in text.splitlines() if not l.startswith("%") for t in l.split()], dtype=np.float64)
    if flat.size % ncols:
        raise MatrixMarketError("entry count %d is not a multiple of %d" % (flat.size, ncols))
    return flat.reshape(-1, ncols)


def coo_to_csr(m, n, rows, cols, vals):
    """Sorted CSR with duplicate (i, j) entries summed: (rowptr int64, col int32, val float64)."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    if rows.size and (rows.min() < 0 or rows.max() >= m or cols.min() < 0 or cols.max() >= n):
        raise MatrixMarketError("index outside the %d x %d matrix" % (m, n))
    key = rows * n + cols
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    if key.size:
        first = np.concatenate([[True], key[1:] != key[:-1]])
        starts = np.flatnonzero(first)
        vals = np.add.reduceat(vals, starts)
        key = key[first]
    r = key // n
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return rowptr, (key - r * n).astype(np.int32), vals


def read(path):
    """-> ("csr", (m, n, rowptr, col, val)) or ("dense", ndarray m x n)."""
    with open(path, "r") as f:
        fmt, field, sym, size = _header(f)
        if fmt == "array":
            if len(size) != 2:
                raise MatrixMarketError("array size line needs 2 integers")
            m, n = size
            if sym != "general":
                raise MatrixMarketError("symmetric array files are not supported")
            data = _numbers(f, 1).ravel()
            if data.size != m * n:
                raise MatrixMarketError("array file holds %d values, expected %d" % (data.size, m * n))
            return "dense", np.asfortranarray(data.reshape((m, n), order="F"))
        if len(size) != 3:
            raise MatrixMarketError("coordinate size line needs 3 integers")
        m, n, nnz = size
        ncols = 2 if field == "pattern" else 3
        ent = _numbers(f, ncols) if nnz else np.zeros((0, ncols))
        if ent.shape[0] != nnz:
            raise MatrixMarketError("coordinate file holds %d entries, header says %d" % (ent.shape[0], nnz))
        rows = ent[:, 0].astype(np.int64) - 1
        cols = ent[:, 1].astype(np.int64) - 1
        vals = np.ones(nnz) if field == "pattern" else ent[:, 2]
        if sym != "general":
            off = rows != cols
            sign = -1.0 if sym == "skew-symmetric" else 1.0
            rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]), np.concatenate([vals, sign * vals[off]]))
        return "csr", (m, n) + coo_to_csr(m, n, rows, cols, vals)


def _column(path):
    with open(path, "r") as f:
        return _numbers(f, 1).ravel()


def read_begjco(stem):
    """CSR matrix from <stem>.info / .beg / .jco / .co (matlab/test/test_MOC.m:94-121) -> (m, n, rowptr, col, val)"""
    n, nnz = (int(x) for x in _column(stem + ".info")[:2])
    beg = _column(stem + ".beg").astype(np.int64)
    jco = _column(stem + ".jco").astype(np.int64)
    co = _column(stem + ".co")
    if beg.size != n + 1 or beg[0] != 1 or beg[-1] != nnz + 1 or np.any(np.diff(beg) < 0):
        raise MatrixMarketError("%s.beg does not hold %d + 1 ascending row starts ending at nnz + 1" % (stem, n))
    if jco.size != nnz or co.size != nnz or (nnz and (jco.min() < 1 or jco.max() > n)):
        raise MatrixMarketError("%s.jco / .co do not hold %d entries with column indices in 1..%d" % (stem, nnz, n))
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(beg))
    return (n, n) + coo_to_csr(n, n, rows, jco - 1, co)


def write_begjco(stem, rowptr, col, val):
    n = len(rowptr) - 1
    with open(stem + ".info", "w") as f:
        f.write("%12d%12d\n" % (n, len(val)))
    np.savetxt(stem + ".beg", np.asarray(rowptr, dtype=np.int64) + 1, fmt="%12d")
    np.savetxt(stem + ".jco", np.asarray(col, dtype=np.int64) + 1, fmt="%12d")
    np.savetxt(stem + ".co", np.asarray(val, dtype=np.float64), fmt="%26.16E")


def _is_begjco(path):
    return not path.endswith(".mtx") and os.path.exists(path + ".beg")


def read_csr(path):
    if _is_begjco(path):
        return read_begjco(path)
    if path.endswith(".co"):  # a diagonal (matlab/test/test_MOC.m:122: M = sparse(1:n, 1:n, Mco))
        d = _column(path)
        n = d.size
        return n, n, np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.int32), d
    kind, data = read(path)
    if kind == "dense":
        m, n = data.shape
        r, c = np.nonzero(data)
        return (m, n) + coo_to_csr(m, n, r, c, data[r, c])
    return data


def read_dense(path):
    """Dense m x n array from either an array file or a (sparse) coordinate file, e.g. the reference's B.mtx."""
    if path.endswith(".co"):  # one column of numbers
        return np.asfortranarray(_column(path)[:, None])
    kind, data = read(path)
    if kind == "dense":
        return data
    m, n, rowptr, col, val = data
    out = np.zeros((m, n), order="F")
    rows = np.repeat(np.arange(m), np.diff(rowptr))
    out[rows, col] = val
    return out


def write_array(path, a, comment=None):
    """`matrix array real general`, column-major, 17 significant digits (what MultiVectorToMatrixMarketFile writes)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        f.write("%%MatrixMarket matrix array real general\n")
        if comment:
            for l in str(comment).splitlines():
                f.write("% " + l + "\n")
        f.write("%d %d\n" % a.shape)
        np.savetxt(f, a.reshape(-1, order="F"), fmt="%.17g")
    os.replace(tmp, path)


def write_csr(path, m, n, rowptr, col, val, comment=None):
    rowptr = np.asarray(rowptr)
    rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(rowptr)) + 1
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        if comment:
            for l in str(comment).splitlines():
                f.write("% " + l + "\n")
        f.write("%d %d %d\n" % (m, n, len(val)))
        np.savetxt(f, np.column_stack([rows, np.asarray(col, dtype=np.int64) + 1, np.asarray(val, dtype=np.float64)]), fmt="%d %d %.17g")
    os.replace(tmp, path)


# ---- parameter files -------------------------------------------------------------------------------------------------------

def _convert(value, typ):
    t = (typ or "string").lower()
    if t in ("int", "unsigned int", "long", "long long", "short"):
        return int(value)
    if t in ("double", "float"):
        return float(value)
    if t == "bool":
        return 1.0 if str(value).strip().lower() in ("true", "1", "yes", "on") else 0.0
    try:
        return float(value)
    except ValueError:
        return value


def _xml_sublist(node):
    out = {}
    for ch in node:
        if ch.tag == "Parameter":
            out[ch.get("name")] = _convert(ch.get("value"), ch.get("type"))
        elif ch.tag == "ParameterList":
            out[ch.get("name")] = _xml_sublist(ch)
    return out


def read_parameters(path, sublist="Lyapunov Solver"):
    """Solver parameters from a Teuchos XML parameter file (the "Lyapunov Solver" sublist, anywhere in the tree; the root list
    itself when there is no such sublist) or from a JSON object.  Values come back as numbers (bools as 0/1)."""
    with open(path, "r") as f:
        text = f.read()
    if text.lstrip().startswith("{"):
        tree = json.loads(text)
    else:
        root = ET.fromstring(text)
        if root.tag != "ParameterList":
            raise ValueError("%s: root element is <%s>, expected <ParameterList>" % (path, root.tag))
        tree = _xml_sublist(root)

    def find(d):
        if sublist in d and isinstance(d[sublist], dict):
            return d[sublist]
        for v in d.values():
            if isinstance(v, dict):
                r = find(v)
                if r is not None:
                    return r
        return None

    sub = find(tree)
    params = sub if sub is not None else tree
    out = {}
    for k, v in params.items():
        if isinstance(v, dict):
            continue
        if isinstance(v, bool):
            v = 1.0 if v else 0.0
        if isinstance(v, (int, float)):
            out[k] = v
    return out
