"""N1: the reference's graph builder (src/step2_graph_builder.py:8-40) as a host-side float64
restatement — a one-off O(S^2) setup step that the reference also runs on the host in float64
(`adj_matrix = torch.tensor(build_graph(df)).float()`, src/main.py:25-26).

  build_adjacency(lat_lon)  [S,2] degrees -> D^-1/2 A_hat D^-1/2  (float64 ndarray)
  build_graph(df)           same signature as the reference: a DataFrame with "Station Name",
                            "Latitude", "Longitude"; rows in first-appearance order (:18).
"""
from __future__ import annotations

import math

import numpy as np

EARTH_RADIUS = 6378137.0          # src/step2_graph_builder.py:9


def mercator(lat_lon: np.ndarray) -> np.ndarray:
    """src/step2_graph_builder.py:8-13 (the reference calls it wgs2utm; it is spherical Web-Mercator)."""
    c = np.asarray(lat_lon, dtype=np.float64)
    out = np.empty_like(c)
    out[:, 0] = EARTH_RADIUS * np.log(np.tan(math.pi / 4 + c[:, 0] * math.pi / 360))
    out[:, 1] = EARTH_RADIUS * (c[:, 1] * math.pi / 180)
    return out


def build_adjacency(lat_lon: np.ndarray) -> np.ndarray:
    p = mercator(lat_lon)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1) / 100000000.0          # :27-30
    with np.errstate(divide="ignore"):
        A_hat = 1.0 / np.sqrt(d2)
    np.fill_diagonal(A_hat, 1.0)                                               # self loops, :24
    dinv = 1.0 / np.sqrt(A_hat.sum(axis=0))                                    # D = diag(colsum), :34,37
    return dinv[:, None] * A_hat * dinv[None, :]                               # :38


def build_graph(df) -> np.ndarray:
    stations = df[["Station Name", "Latitude", "Longitude"]].drop_duplicates()  # :18
    return build_adjacency(stations[["Latitude", "Longitude"]].values)


# ---- large graphs: k-NN sparsification + CSR (BASELINE config 5) -------------------

def synthetic_station_coords(n: int, seed: int = 7) -> np.ndarray:
    """n points uniform over the span of the reference's station list (lat 50..52, lon -114..-110)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(50.0, 52.0, n), rng.uniform(-114.0, -110.0, n)], axis=1)


def build_knn_adjacency(lat_lon: np.ndarray, k: int = 8):
    """The reference's inverse-distance weights (src/step2_graph_builder.py:27-31) kept only on the symmetric
    k-nearest-neighbour edges, self loops 1 (:24), then D^-1/2 A_hat D^-1/2 (:34-38).  Returns
    (rowptr int32[S+1], col int32[nnz], val float64[nnz]) with columns ascending inside each row."""
    p = mercator(lat_lon)
    S = p.shape[0]
    k = min(k, S - 1)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1) / 100000000.0
    np.fill_diagonal(d2, np.inf)
    nbr = np.argpartition(d2, k - 1, axis=1)[:, :k] if k > 0 else np.empty((S, 0), dtype=np.int64)
    keep = np.zeros((S, S), dtype=bool)
    keep[np.repeat(np.arange(S), k), nbr.ravel()] = True
    keep |= keep.T                                                             # symmetric k-NN
    with np.errstate(divide="ignore"):
        A_hat = np.where(keep, 1.0 / np.sqrt(d2), 0.0)
    np.fill_diagonal(A_hat, 1.0)
    dinv = 1.0 / np.sqrt(A_hat.sum(axis=0))
    A = dinv[:, None] * A_hat * dinv[None, :]
    mask = keep | np.eye(S, dtype=bool)
    rowptr = np.concatenate([[0], np.cumsum(mask.sum(axis=1))]).astype(np.int32)
    rows, cols = np.nonzero(mask)                                              # row-major: ascending columns
    return rowptr, cols.astype(np.int32), A[rows, cols]


class CsrAdjacency:
    """A sparse adjacency in the layout the C ABI takes for WGNN_ADJ_CSR (include/windgnn.h): one int32
    buffer holding A and A^T, each as rowptr | col | val(fp32 bits)."""

    def __init__(self, rowptr, col, val, S: int | None = None):
        import torch
        rowptr = np.asarray(rowptr, dtype=np.int64)
        col = np.asarray(col, dtype=np.int64)
        val = np.asarray(val, dtype=np.float32)
        self.S = int(S if S is not None else len(rowptr) - 1)
        self.nnz = int(len(col))
        if len(rowptr) != self.S + 1 or rowptr[0] != 0 or rowptr[-1] != self.nnz or len(val) != self.nnz:
            raise ValueError("CsrAdjacency: inconsistent rowptr / col / val")
        if self.nnz and (col.min() < 0 or col.max() >= self.S):
            raise ValueError("CsrAdjacency: column index out of range")
        row = np.repeat(np.arange(self.S), np.diff(rowptr))
        order = np.lexsort((col, row))                                         # ascending columns inside a row
        row, col, val = row[order], col[order], val[order]
        t_order = np.lexsort((row, col))                                       # transpose: sort by (col, row)
        t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=self.S))])
        words = [rowptr.astype(np.int32), col.astype(np.int32), val.view(np.int32),
                 t_rowptr.astype(np.int32), row[t_order].astype(np.int32), val[t_order].view(np.int32)]
        self.blob = torch.from_numpy(np.concatenate(words))
        self._rows, self._cols, self._vals = row, col, val

    @classmethod
    def from_dense(cls, A, tol: float = 0.0):
        A = np.asarray(A.detach().cpu() if hasattr(A, "detach") else A, dtype=np.float64)
        mask = np.abs(A) > tol
        rowptr = np.concatenate([[0], np.cumsum(mask.sum(axis=1))])
        rows, cols = np.nonzero(mask)
        return cls(rowptr, cols, A[rows, cols], S=A.shape[0])

    def to(self, device):
        self.blob = self.blob.to(device)
        return self

    def dense(self):
        """fp32 dense copy (for the CPU oracle in tests)."""
        import torch
        A = torch.zeros(self.S, self.S, dtype=torch.float32)
        A[torch.from_numpy(self._rows), torch.from_numpy(self._cols)] = torch.from_numpy(self._vals)
        return A
