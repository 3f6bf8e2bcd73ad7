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
