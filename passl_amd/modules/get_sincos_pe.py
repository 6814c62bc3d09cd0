"""Fixed 2-D sin-cos position table of MAE (function names and argument order follow
passl_v110/modules/get_sincos_pe.py:18-75, so code written against the reference finds them).

Token (h, w) of a g x g patch grid gets a D-vector made of four D/4 blocks,
    [ sin(w * f), cos(w * f), sin(h * f), cos(h * f) ],   f_i = 10000 ** (-i / (D/4)),  i < D/4
— the column coordinate comes FIRST (the reference builds its mesh with w first); an optional leading
all-zero row is the class token's slot.  float64 throughout, like the reference."""
import numpy as np


def _frequencies(n):
    return np.power(10000.0, -np.arange(n, dtype=np.float64) / n)


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """[sin | cos] of pos x frequencies: positions (any shape, flattened) -> [M, embed_dim]."""
    if embed_dim % 2:
        raise AssertionError('embed_dim must be even')
    phase = np.outer(np.asarray(pos, dtype=np.float64).ravel(), _frequencies(embed_dim // 2))
    return np.hstack((np.sin(phase), np.cos(phase)))


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    """grid[0] / grid[1] = the two coordinate planes; each contributes embed_dim/2 columns."""
    if embed_dim % 2:
        raise AssertionError('embed_dim must be even')
    half = embed_dim // 2
    return np.hstack([get_1d_sincos_pos_embed_from_grid(half, plane) for plane in (grid[0], grid[1])])


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    coords = np.arange(grid_size, dtype=np.float32)
    cols, rows = np.meshgrid(coords, coords)             # cols[h, w] = w, rows[h, w] = h
    table = get_2d_sincos_pos_embed_from_grid(embed_dim, np.stack((cols, rows))[:, None])   # w first
    if cls_token:
        table = np.vstack((np.zeros((1, embed_dim)), table))
    return table
