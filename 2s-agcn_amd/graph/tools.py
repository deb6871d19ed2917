"""Skeleton-graph adjacency construction.

Behaviour follows the reference ``graph/tools.py:4-27`` (edge2mat,
normalize_digraph, get_spatial_graph): ``A = stack(I, norm(inward), norm(outward))``
with ``edge2mat`` setting ``A[j, i] = 1`` for an edge ``(i, j)`` and the
normalisation dividing every COLUMN by its sum (columns with sum 0 stay 0).
"""
import numpy as np


def edge_matrix(edges, num_node):
    a = np.zeros((num_node, num_node), dtype=np.float64)
    for i, j in edges:
        a[j, i] = 1.0
    return a


def column_normalize(a):
    col = a.sum(axis=0)
    scale = np.zeros_like(col)
    nz = col > 0
    scale[nz] = 1.0 / col[nz]
    return a * scale[None, :]


def spatial_graph(num_node, self_link, inward, outward):
    return np.stack((edge_matrix(self_link, num_node),
                     column_normalize(edge_matrix(inward, num_node)),
                     column_normalize(edge_matrix(outward, num_node))))
